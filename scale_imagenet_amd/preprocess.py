"""GPU side of the reference's eval input transform (utils/preprocess.py:104-108, main.py:208):

    transforms.Resize(256) -> transforms.CenterCrop(224) -> transforms.ToTensor() -> transforms.Normalize(mean, std)

``resize_center_crop_u8`` does the first two on decoded uint8 HWC images already on the device
(libttnet: ttnet_resize_center_crop_u8, csrc/preproc.hip); the last two are fused into the stem by
``model.forward_u8``.  JPEG decoding stays on the host (out of scope).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def resize_center_crop_u8(x_u8: torch.Tensor, resize: int = 256, crop: int = 224) -> torch.Tensor:
    """uint8 HIP tensor [N,H,W,3] (one image size per call, as a decoder batch delivers) ->
    uint8 [N,crop,crop,3]."""
    if (not x_u8.is_cuda) or x_u8.dtype != torch.uint8 or x_u8.dim() != 4 or x_u8.shape[3] != 3:
        raise RuntimeError(f"expected a uint8 HIP tensor [N,H,W,3], got {x_u8.dtype} {tuple(x_u8.shape)} on {x_u8.device}")
    x_u8 = x_u8.contiguous()
    n, h, w, _ = x_u8.shape
    out = torch.empty((n, crop, crop, 3), device=x_u8.device, dtype=torch.uint8)
    with torch.cuda.device(x_u8.device):
        stream = torch.cuda.current_stream(x_u8.device).cuda_stream
        _lib.check(_lib.load().ttnet_resize_center_crop_u8(C.c_void_p(x_u8.data_ptr()), n, h, w, int(resize), int(crop),
                                                           C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
    return out


def imgnet_eval_forward(model, x_u8: torch.Tensor, lane: int = 0) -> torch.Tensor:
    """``model(imgnet_transform(False)(image))`` for a batch of decoded images of one size."""
    return model.forward_u8(resize_center_crop_u8(x_u8), lane=lane)
