"""nn.Module mirror of the reference's TTNet ImageNet classifiers, backed by libttnet.so.

Drop-in boundary: same class names, same constructor arguments
(``args.nfilter / tfilter / layers / groups``), same ``features`` container and the same
``state_dict`` keys, shapes, dtypes and order as

  * ``TT_vf_19lv3_imgnet_small``   models/TT_general_imagenet_v2_small.py:151-207
  * ``TT_vf_19lv3_imgnet_xsmall``  models/TT_general_imagenet_v2_xsmall.py:151-207
  * ``TT_vf_19lv3_imgnet``         models/TT_general_imagenet_v2.py:139-196

so ``main.py:148`` (construct), ``:222`` (``load_state_dict``, strict), ``:251`` (``eval()``)
and ``:261`` (``model(inputs)``) run unchanged.  The sub-modules below only hold parameters
and buffers under the reference's names; they compute nothing.  ``forward`` hands raw device
pointers to the C ABI (include/ttnet.h); there is no eager / CPU fallback -- a CPU tensor,
training mode or a missing library raise.
"""
from __future__ import annotations

import ctypes as C
import threading
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .spec import (BlockTTSpec, MultiHeadSpec, VAlexSpec, VariantSpec, make_spec, state_dict_layout,
                   valexnet_layout)

_DT = {torch.float32: _lib.TTNET_F32, torch.int64: _lib.TTNET_I64}


class _Holder(nn.Module):
    """A module that only owns state; the compute lives in libttnet."""

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise NotImplementedError(
            f"{type(self).__name__} holds parameters only; call the model (its forward runs the "
            "HIP path end to end)")


class Binarize01Act(_Holder):
    """State of both activation flavours (netbin.py:184-239, TT_FHE_SMALL.py:176-248):
    one float32 scalar buffer ``grad_scale`` that eval-mode never reads."""

    def __init__(self, T: float = 0.0, grad_scale: float = 1.0):
        super().__init__()
        self.T = T
        self.register_buffer("grad_scale", torch.tensor(float(grad_scale), dtype=torch.float32))


class Block_TT(_Holder):
    """State of models/TT_FHE_SMALL.py:278-305."""

    def __init__(self, b: BlockTTSpec):
        super().__init__()
        mid = 8 * b.in_planes
        self.conv1 = nn.Conv2d(b.in_planes, mid, kernel_size=(b.kh, b.kw), stride=b.stride, padding=0,
                               groups=b.groups, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, b.out_planes, kernel_size=1, stride=1, padding=0, groups=b.groups, bias=False)
        self.bn2 = nn.BatchNorm2d(b.out_planes)
        self.act = Binarize01Act(T=0.0)
        self.last = b.last
        self.groupsici = b.groups
        self.in_planes = b.in_planes


class Block_resnet_multihead_general_BN_vf_imgnet_v2small(_Holder):
    """State of models/TT_general_imagenet_v2_small.py:24-76 (registration order
    Block_conv1, Block_conv2, Block_conv3, Block_conv4, act, Block_convf)."""

    def __init__(self, m: MultiHeadSpec):
        super().__init__()
        self.Block_conv1 = Block_TT(m.conv1)
        self.Block_conv2 = Block_TT(m.conv2)
        self.Block_conv3 = Block_TT(m.conv3)
        self.Block_conv4 = nn.AvgPool2d(2) if m.stride == 2 else None
        self.act = Binarize01Act()
        self.Block_convf = Block_TT(m.convf)
        self.stride = m.stride
        self.last = m.last
        self.cpt = 4


class Polynome_ACT(_Holder):
    """models/TT_general_imagenet_v2_small.py:209-215 (no state)."""


class Classifier_scale(_Holder):
    """State of models/TT_general_imagenet_v2_small.py:217-227."""

    def __init__(self, fcsize: int, out_planes: int, inter: int = 1000):
        super().__init__()
        self.lin1 = nn.Linear(fcsize, inter, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.BN2 = nn.BatchNorm1d(inter)
        self.lin2 = nn.Linear(inter, 1000, bias=True)
        self.Polynome_ACT = Polynome_ACT()


class Flatten(_Holder):
    """models/model_utils/utils.py:261-267 (no state)."""


class _Plan:
    """One libttnet plan: a device, a max batch and the state it was finalized with."""

    def __init__(self, spec: VariantSpec, args, device_index: int, max_batch: int):
        lib = _lib.load()
        self.lib = lib
        self.device_index = device_index
        self.max_batch = max_batch
        self.handle = C.c_void_p()
        desc = _lib.NetDesc(_lib.VARIANTS[spec.variant], int(getattr(args, "nfilter", 8)),
                            int(getattr(args, "tfilter", 8)), int(getattr(args, "layers", 1)),
                            spec.image_hw[0], spec.image_hw[1], int(max_batch), 0)
        _lib.check(lib.ttnet_plan_create(C.byref(desc), device_index, C.byref(self.handle)))
        self.signature = None

    def load(self, state: "OrderedDict[str, torch.Tensor]", stream: int):
        for key, t in state.items():
            t = t.detach()
            if t.dtype not in _DT:
                raise RuntimeError(f"{key}: unsupported dtype {t.dtype}")
            t = t.contiguous()
            on_dev = 1 if (t.is_cuda and t.device.index == self.device_index) else 0
            if t.is_cuda and not on_dev:
                t = t.cpu()
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            _lib.check(self.lib.ttnet_plan_set_tensor(self.handle, key.encode(), C.c_void_p(t.data_ptr()), shape,
                                                      t.dim(), _DT[t.dtype], on_dev))
        _lib.check(self.lib.ttnet_plan_finalize(self.handle, C.c_void_p(stream)))

    def set_lanes(self, lanes: int):
        _lib.check(self.lib.ttnet_plan_set_lanes(self.handle, int(lanes)))

    def query(self, what: str) -> int:
        out = C.c_int64()
        _lib.check(self.lib.ttnet_plan_query(self.handle, what.encode(), C.byref(out)))
        return out.value

    def close(self):
        if self.handle:
            self.lib.ttnet_plan_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass


class _TTNetBase(nn.Module):
    VARIANT = "small"
    CLASS2NAME = tuple(map(str, range(10)))     # kept from the reference class
    DEFAULT_MAX_BATCH = 256

    def __init__(self, args):
        super().__init__()
        self.args = args
        if self.VARIANT == "valexnet":
            self._init_valexnet()
            return
        spec = make_spec(self.VARIANT, int(args.nfilter), int(args.tfilter), int(args.layers))
        self.spec = spec
        layers = [nn.AvgPool2d(2),
                  nn.Conv2d(3, spec.p, kernel_size=7, stride=2, padding=3, groups=1, bias=False),
                  nn.BatchNorm2d(spec.p),
                  Binarize01Act()]
        for m in spec.blocks:
            layers.append(Block_resnet_multihead_general_BN_vf_imgnet_v2small(m))
        layers.append(nn.AvgPool2d(2))
        layers.append(Flatten())
        layers.append(Classifier_scale(spec.fcsize, 10, 1000))
        self.features = nn.Sequential(*layers)
        # the reference's constructor dry-runs the net once in training mode (:199-207), which
        # leaves every BatchNorm at num_batches_tracked == 1
        for mod in self.modules():
            if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm1d)):
                mod.num_batches_tracked.fill_(1)
        self._plans: Dict[int, _Plan] = {}
        self.__dict__["_plans_lock"] = threading.Lock()
        layout = state_dict_layout(spec)
        mine = self.state_dict()
        assert list(mine.keys()) == list(layout.keys()), "state_dict layout drifted from spec"

    def _init_valexnet(self):
        """State of TT_FHE_XSMALL_vAlexnet (models/TT_FHE_XSMALL_vAlexnet.py:585-660): the stem
        conv is registered as ``VGG_Model16_0`` and again inside ``features`` (57 keys)."""
        spec = VAlexSpec()
        self.spec = spec
        self.VGG_Model16_0 = nn.Conv2d(3, 64, 3, padding=1)
        self.VGG_Model16_1 = nn.ReLU(inplace=True)
        blk = _Holder()
        blk.pad0 = nn.ZeroPad2d((1, 0, 1, 0))
        blk.Block_conv1 = Block_TT(spec.conv1)
        blk.Block_conv2 = Block_TT(spec.conv2)
        blk.Block_conv3 = Block_TT(spec.conv3)
        head = _Holder()
        head.lin1 = nn.Linear(spec.fcsize, spec.inter, bias=False)
        head.BN2 = nn.BatchNorm1d(spec.inter)
        head.lin2 = nn.Linear(spec.inter, spec.n_classes, bias=True)
        self.features = nn.Sequential(self.VGG_Model16_0, self.VGG_Model16_1, nn.BatchNorm2d(64), nn.MaxPool2d(3),
                                      Binarize01Act(T=0.0), blk, Flatten(), head)
        for mod in self.modules():
            if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm1d)):
                mod.num_batches_tracked.fill_(1)
        self._plans = {}
        self.__dict__["_plans_lock"] = threading.Lock()
        assert list(self.state_dict().keys()) == list(valexnet_layout(spec).keys()), "state_dict layout drifted"

    # -- plan management ------------------------------------------------------------------
    def _state_signature(self):
        """(storage address, in-place version) of every state tensor.  ``state_dict()`` itself
        costs ~0.5 ms of Python per call -- more than the whole forward at batch 256 -- so the
        tensor list is cached; ``.to()`` / ``.cuda()`` / ``load_state_dict`` and in-place updates
        are seen through the addresses and version counters, and the list itself is rebuilt by
        ``_apply`` / ``load_state_dict`` / ``refresh_state()`` (call the latter after assigning a
        new Parameter object to a sub-module by hand)."""
        src = self.__dict__.get("_sig_source")
        if src is not None:                  # a DataParallel replica: see _replicate_for_data_parallel
            return src._state_signature()
        ts = self.__dict__.get("_sig_tensors")
        if ts is None:
            ts = list(self.state_dict().values())
            self.__dict__["_sig_tensors"] = ts
        return tuple((t.data_ptr(), t._version) for t in ts)

    def refresh_state(self):
        self.__dict__["_sig_tensors"] = None
        return self

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.__dict__["_sig_tensors"] = None
        return out

    def __getstate__(self):
        """copy.deepcopy / pickling: plans (device handles), their lock and the signature cache stay behind."""
        d = dict(self.__dict__)
        d["_plans"] = {}
        for k in ("_plans_lock", "_sig_tensors", "_sig_source"):
            d.pop(k, None)
        return d

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__["_plans_lock"] = threading.Lock()
        self.__dict__["_sig_tensors"] = None

    def _replicate_for_data_parallel(self):
        """``nn.DataParallel`` over several devices (main.py:192 wraps over all visible GPUs): ``replicate`` makes a
        shallow copy of the module per device and per forward, whose parameters are fresh broadcast copies.  A
        replica therefore (1) shares this module's per-device plan cache -- one plan per device, kept across
        forwards, creation and reload serialised by a lock shared with the original (``parallel_apply`` runs the
        replicas on threads) -- and (2) takes its state signature from the ORIGINAL's tensors: its own copies are
        new tensors on every forward, so their addresses say nothing, while the original's (address, version)
        pairs change exactly when the weights do.  A replica's plan is then (re)loaded from the replica's own,
        device-local state_dict only when the weights changed: truth tables are not rebuilt per forward (10.5 ms)."""
        replica = super()._replicate_for_data_parallel()
        replica.__dict__["_sig_source"] = self.__dict__.get("_sig_source") or self
        replica.__dict__["_sig_tensors"] = None
        return replica

    def _plan_for(self, device: torch.device, n: int) -> _Plan:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        with self.__dict__["_plans_lock"]:
            plan = self._plans.get(idx)
            if plan is not None and n > plan.max_batch:
                plan.close()
                plan = None
            if plan is None:
                plan = _Plan(self.spec, self.args, idx, max(n, self.DEFAULT_MAX_BATCH))
                if self.__dict__.get("_lanes", 1) > 1:
                    plan.set_lanes(self._lanes)
                self._apply_input_norm(plan)
                self._plans[idx] = plan
            sig = self._state_signature()
            if plan.signature != sig:
                plan.load(self.state_dict(), torch.cuda.current_stream(device).cuda_stream)
                plan.signature = sig
        return plan

    def set_lanes(self, lanes: int):
        """Activation workspaces for ``lanes`` batches in flight (``model(x, lane=k)``, each on
        its own stream; see ttnet_forward_lane).  Weights and truth tables are shared."""
        self.__dict__["_lanes"] = int(lanes)
        for plan in self._plans.values():
            plan.set_lanes(int(lanes))
        return self

    def reserve(self, max_batch: int):
        """Size the activation workspace up front (otherwise it grows on demand)."""
        self.DEFAULT_MAX_BATCH = int(max_batch)
        return self

    # -- the hot path ------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, lane: int = 0) -> torch.Tensor:
        """``self.features(x)`` of the reference (netbin.py:703-708), eval mode.  ``lane`` picks
        the activation workspace (``set_lanes``) when several batches are kept in flight.
        A value outside the range of the split operands makes the NEXT call raise (sticky flag): after the last
        batch of a loop, synchronise and call ``check_range()`` (``evaluate.evaluate`` does)."""
        if self.training:
            raise RuntimeError("the HIP path implements eval-mode inference only: call model.eval() "
                               "(main.py:251); training is out of scope")
        if not x.is_cuda:
            raise RuntimeError("scale_imagenet_amd has no CPU path: move the model and the input to a "
                               "HIP device (the CPU restatement lives in oracle/ and is test-only)")
        h, w = self.spec.image_hw
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, h, w):
            raise RuntimeError(f"expected input [N,3,{h},{w}], got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            raise RuntimeError(f"expected float32 input, got {x.dtype}")
        x = x.contiguous()
        if x.data_ptr() % 16:                 # a view into a larger buffer: the stem wants 16-byte alignment (ttnet.h)
            x = x.clone()
        n = x.shape[0]
        plan = self._plan_for(x.device, n)
        out = torch.empty((n, self.spec.n_classes), device=x.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(plan.lib.ttnet_forward_lane(plan.handle, int(lane), C.c_void_p(x.data_ptr()), n,
                                               C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    def forward_u8(self, x_u8: torch.Tensor, lane: int = 0) -> torch.Tensor:
        """Forward from the decoder's uint8 HWC images ``[N,H,W,3]``: ToTensor + Normalize
        (utils/preprocess.py:104-108) are applied inside the stem kernel."""
        if self.training:
            raise RuntimeError("the HIP path implements eval-mode inference only: call model.eval()")
        h, w = self.spec.image_hw
        if (not x_u8.is_cuda) or x_u8.dtype != torch.uint8 or x_u8.dim() != 4 or tuple(x_u8.shape[1:]) != (h, w, 3):
            raise RuntimeError(f"expected a uint8 HIP tensor [N,{h},{w},3], got {x_u8.dtype} {tuple(x_u8.shape)} on {x_u8.device}")
        x_u8 = x_u8.contiguous()
        if x_u8.data_ptr() % 4:
            x_u8 = x_u8.clone()
        n = x_u8.shape[0]
        plan = self._plan_for(x_u8.device, n)
        out = torch.empty((n, self.spec.n_classes), device=x_u8.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(x_u8.device).cuda_stream
        _lib.check(plan.lib.ttnet_forward_u8(plan.handle, int(lane), C.c_void_p(x_u8.data_ptr()), n,
                                             C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    def export_truth_tables(self, block_name: str, out_dir: str, block: int = 0, sub_block: int = 0, filters=None,
                            max_expr_bits: int = 9):
        """Write the reference's truth-table files (CSV, DNF / CNF, SAT form) for one
        ``Block_TT`` from the table the plan built on the GPU.  See ``scale_imagenet_amd.export``."""
        from . import export
        return export.export_block(self.get_table(block_name), out_dir, block, sub_block, filters, max_expr_bits)

    def set_input_norm(self, mean, std):
        """Normalisation constants of ``forward_u8`` (default: the ImageNet ones)."""
        self.__dict__["_input_norm"] = (tuple(float(v) for v in mean), tuple(float(v) for v in std))
        for plan in self._plans.values():
            self._apply_input_norm(plan)
        return self

    def _apply_input_norm(self, plan):
        norm = self.__dict__.get("_input_norm")
        if norm is not None:
            _lib.check(plan.lib.ttnet_plan_set_input_norm(plan.handle, (C.c_float * 3)(*norm[0]), (C.c_float * 3)(*norm[1])))

    # -- parity taps (replace Block_TT.input_layer / output_layer, TT_FHE_SMALL.py:310,319) --
    def forward_from_stem_bits(self, rows: torch.Tensor) -> torch.Tensor:
        """Gate path + head from row-packed stem bits (uint64 viewed as int64 [N,p,56])."""
        assert rows.is_cuda and rows.dtype == torch.int64 and rows.is_contiguous()
        n = rows.shape[0]
        plan = self._plan_for(rows.device, n)
        out = torch.empty((n, self.spec.n_classes), device=rows.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(rows.device).cuda_stream
        _lib.check(plan.lib.ttnet_forward_from_stem_bits(plan.handle, C.c_void_p(rows.data_ptr()), n,
                                                         C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    def check_range(self):
        """Raise if a forward since the last check left the range of the fp16 x 2 split operands (|activation| >= 4094:
        include/ttnet.h, TTNET_E_RANGE).  The kernels raise a sticky flag; the C ABI reports it on the NEXT call of the
        plan, so a loop that only calls ``forward`` would miss an overflow in its last batch: call this after the
        synchronisation that ends the loop (``evaluate`` does).  Synchronises the device."""
        for plan in self._plans.values():
            if plan.query("range_overflow"):
                raise RuntimeError("ttnet: an activation left the range of the split operands (|value| >= 4094); "
                                   "the logits of that batch are invalid (TTNET_E_RANGE)")

    def _any_plan(self) -> _Plan:
        if not self._plans:
            raise RuntimeError("no forward has run yet")
        return next(iter(self._plans.values()))

    def read_stage(self, stage: str, n: int) -> np.ndarray:
        plan = self._any_plan()
        if stage == "flatten":
            out = np.empty((n, self.spec.fcsize), dtype=np.float32)
        else:
            c, hh = self._stage_shape(stage)
            out = np.empty((n, c, hh), dtype=np.uint64)
        stream = torch.cuda.current_stream(torch.device("cuda", plan.device_index)).cuda_stream
        _lib.check(plan.lib.ttnet_read_stage(plan.handle, stage.encode(), n, out.ctypes.data_as(C.c_void_p),
                                             out.nbytes, 0, C.c_void_p(stream)))
        return out

    def _stage_shape(self, stage: str):
        if self.VARIANT == "valexnet":
            return {"features.4": (64, 10), "features.5": (256, 11)}[stage]
        if stage == "features.3":
            return self.spec.p, 56
        for b in self.spec.blocks:
            if stage.startswith(b.name + ".out"):
                return b.in_planes, b.out_hw[0]
            if stage == b.name:
                return b.convf.out_planes, b.out_hw[0]
        raise KeyError(stage)

    def get_table(self, name: str) -> np.ndarray:
        """Truth table of Block_TT ``name`` in the reference's canonical order
        ([groups, 2^n, cout_g]; uint8 bits, float32 for the last block)."""
        plan = self._any_plan()
        b = {x.name: x for x in self.spec.block_tts()}[name]
        shape = (b.groups, 1 << b.fan_in_bits, b.cout_g)
        out = np.empty(shape, dtype=np.float32 if b.last else np.uint8)
        _lib.check(plan.lib.ttnet_plan_get_table(plan.handle, name.encode(), out.ctypes.data_as(C.c_void_p),
                                                 out.nbytes))
        return out

    def set_table(self, name: str, table: np.ndarray):
        plan = self._any_plan()
        t = np.ascontiguousarray(table)
        _lib.check(plan.lib.ttnet_plan_set_table(plan.handle, name.encode(), t.ctypes.data_as(C.c_void_p), t.nbytes))

    def near_ties(self) -> Dict[str, int]:
        plan = self._any_plan()
        return {b.name: plan.query("near_ties:" + b.name) for b in self.spec.block_tts()}

    def set_profiling(self, enabled: bool):
        for plan in self._plans.values():
            _lib.check(plan.lib.ttnet_plan_set_profiling(plan.handle, int(enabled)))

    def last_timings(self) -> "OrderedDict[str, float]":
        plan = self._any_plan()
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        k = _lib.check(plan.lib.ttnet_plan_last_timings(plan.handle, names, ms, cap))
        out: "OrderedDict[str, float]" = OrderedDict()
        for i in range(k):
            out[names[i].decode()] = float(ms[i])
        return out

    # state_dict compatibility: accept DataParallel / DDP checkpoints (main.py:181-192, :222)
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        if state_dict and all(k.startswith("module.") for k in state_dict):
            state_dict = OrderedDict((k[len("module."):], v) for k, v in state_dict.items())
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.__dict__["_sig_tensors"] = None
        return out


class TT_vf_19lv3_imgnet_small(_TTNetBase):
    VARIANT = "small"


class TT_vf_19lv3_imgnet_xsmall(_TTNetBase):
    VARIANT = "xsmall"


class TT_vf_19lv3_imgnet(_TTNetBase):
    VARIANT = "full"


class TT_FHE_XSMALL_vAlexnet(_TTNetBase):
    """models/TT_FHE_XSMALL_vAlexnet.py:585 (CIFAR 32x32, 10 classes).  Unlike the reference it
    does not fetch pretrained VGG16 weights: the stem conv is an ordinary parameter, loaded from
    the checkpoint like the rest."""
    VARIANT = "valexnet"
