"""Device-side front-end of the hot path: the integer rules between the collator / processor and the first
GEMM as HIP kernels (csrc/frontend.hip), so that a forward pass makes no host round trip.

  * ``idefics_image_attention_mask`` — what ``processor.prepare_input`` hands the model as ``image_attention_mask``
    (ref:icv_src/icv_datamodule.py:80-124 -> hf:idefics/processing_idefics.py:89-133), from ``input_ids`` on the device;
  * ``idefics2_patch_front``         — padding-image flags, patch validity and NaViT position ids
    (hf:idefics2/modeling_idefics2.py:831-855, :136-170);
  * ``merge_image_rows_``            — Idefics2 ``inputs_merger`` (hf:idefics2/modeling_idefics2.py:789-815) without ``nonzero()``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import check
from .ops import _p, _stream


def idefics_image_attention_mask(input_ids: torch.Tensor, image_token_id: int, eod_token_id: int, n_images: int,
                                 dtype: torch.dtype = torch.int32) -> torch.Tensor:
    """(B, S) int64 ids on the GPU -> (B, S, n_images) one-hot rows: token t attends the most recent <image> at or before t,
    nothing before the first image or between an end-of-document token and the next image.  int32 is what the engine's
    attention kernel reads; pass ``dtype=torch.long`` for the processor's own dtype."""
    assert input_ids.is_cuda and input_ids.dtype == torch.int64 and input_ids.dim() == 2
    ids = input_ids.contiguous()
    B, S = ids.shape
    out = torch.empty((B, S, n_images), dtype=torch.int32, device=ids.device)
    check(_lib.lib().licv_idefics_image_attention_mask(_p(ids), _p(out), B, S, n_images, int(image_token_id), int(eod_token_id), _stream(ids)))
    return out if dtype == torch.int32 else out.to(dtype)


_bounds: dict = {}


def navit_boundaries(n_side: int, device) -> torch.Tensor:
    """The fp32 bucket boundaries exactly as the HF module builds them (torch.arange on the host, uploaded once)."""
    key = (n_side, str(device))
    if key not in _bounds:
        _bounds[key] = torch.arange(1 / n_side, 1.0, 1 / n_side).to(device=device, dtype=torch.float32).contiguous()
    return _bounds[key]


def idefics2_patch_front(pixel_values: torch.Tensor, pixel_attention_mask: Optional[torch.Tensor], patch: int, n_side: int):
    """pixel_values (n, 3, H, W) bf16, pixel_attention_mask (n, H, W) bool or None ->
    (real (n,) int32, patch_valid (n, gh*gw) int32, position_ids (n, gh*gw) int64), all on the device."""
    assert pixel_values.is_cuda and pixel_values.dtype == torch.bfloat16 and pixel_values.is_contiguous() and pixel_values.dim() == 4
    n, _, H, W = pixel_values.shape
    gh, gw = H // patch, W // patch
    dev = pixel_values.device
    pam = None
    if pixel_attention_mask is not None:
        pam = pixel_attention_mask.to(device=dev, dtype=torch.bool).reshape(n, H, W).contiguous().view(torch.uint8)
    real = torch.empty((n,), dtype=torch.int32, device=dev)
    valid = torch.empty((n, gh * gw), dtype=torch.int32, device=dev)
    pos = torch.empty((n, gh * gw), dtype=torch.int64, device=dev)
    check(_lib.lib().licv_idefics2_patch_front(_p(pixel_values), _p(pam), _p(navit_boundaries(n_side, dev)), _p(real), _p(valid), _p(pos),
                                               n, H, W, patch, n_side, _stream(pixel_values)))
    return real, valid, pos


def merge_image_rows_(h: torch.Tensor, input_ids: torch.Tensor, image_rows: torch.Tensor, image_token_id: int) -> torch.Tensor:
    """In place: rows of h (M, dim) bf16 whose token is <image> take the image rows in order.  Returns the device-side count of
    <image> tokens (1-element int32 tensor; reading it is the caller's choice — it costs a sync)."""
    assert h.is_cuda and h.dtype == torch.bfloat16 and h.is_contiguous() and image_rows.dtype == torch.bfloat16 and image_rows.is_contiguous()
    ids = input_ids.reshape(-1).contiguous()
    M, dim = h.shape
    assert ids.numel() == M and ids.dtype == torch.int64
    scratch = torch.empty((M + 1,), dtype=torch.int32, device=h.device)
    check(_lib.lib().licv_merge_image_rows(_p(h), _p(ids), _p(image_rows), _p(scratch), C.c_void_p(scratch.data_ptr() + 4 * M), M, dim,
                                           image_rows.shape[0], int(image_token_id), _stream(h)))
    return scratch[M:]


# ------------------------------------------------------------------------------------------ image input (uint8 -> normalised bf16)
IDEFICS_MEAN = (0.48145466, 0.4578275, 0.40821073)        # hf:idefics/image_processing_idefics.py IDEFICS_STANDARD_MEAN / _STD
IDEFICS_STD = (0.26862954, 0.26130258, 0.27577711)
IDEFICS2_MEAN = IDEFICS2_STD = (0.5, 0.5, 0.5)              # hf:idefics2 image processor (IMAGENET_STANDARD_MEAN / _STD)


def preprocess_images(u8: torch.Tensor, mean=IDEFICS_MEAN, std=IDEFICS_STD, rescale: float = 1 / 255,
                      valid_hw: Optional[torch.Tensor] = None, want_mask: bool = False, out: Optional[torch.Tensor] = None,
                      mask_out: Optional[torch.Tensor] = None):
    """u8 (n, H, W, 3) uint8 on the GPU -> (pixel_values (n, 3, H, W) bf16, pixel_attention_mask (n, H, W) bool or None): the
    rescale + normalise + channels-first of the HF image processors (and, with `valid_hw` (n, 2) int32 = real height / width of each
    image inside its padded H x W, Idefics2's zero padding and mask), as one HIP kernel (csrc/frontend.hip).  Resizing stays where
    the reference has it: on the host, in the dataset workers."""
    assert u8.is_cuda and u8.dtype == torch.uint8 and u8.dim() == 4 and u8.shape[-1] == 3 and u8.is_contiguous()
    n, H, W, _ = u8.shape
    dev = u8.device
    if out is None:
        out = torch.empty((n, 3, H, W), dtype=torch.bfloat16, device=dev)
    m = None
    if want_mask or mask_out is not None:
        m = mask_out if mask_out is not None else torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    hw = None
    if valid_hw is not None:
        hw = valid_hw.to(device=dev, dtype=torch.int32).contiguous()
        assert hw.shape == (n, 2)
    mean3, std3 = (C.c_float * 3)(*[float(x) for x in mean]), (C.c_float * 3)(*[float(x) for x in std])
    check(_lib.lib().licv_preprocess_images(_p(u8), _p(hw), _p(out), _p(m), n, H, W, C.c_double(float(rescale)), mean3, std3, _stream(u8)))
    return out, (m.view(torch.bool) if m is not None else None)
