"""Oracle (test infrastructure): the processor-side integer rules of the path, restated as plain loops.

``image_attention_mask`` follows hf:idefics/processing_idefics.py:89-110 (image_attention_mask_for_packed_input_ids_pt, the
forward scan) and :66-79 (incremental_to_binary_attention_mask) token by token — what ``processor.prepare_input`` hands the
reference's collator (ref:icv_src/icv_datamodule.py:80-124).  The Idefics2 patch mask / NaViT position ids live in
``oracle.idefics2_ref`` (patch_mask_from_pixels, navit_position_ids + bucketize, as the vision tower applies them).
"""
import torch


def image_attention_mask(input_ids: torch.Tensor, image_token_id: int, eod_token_id: int, n_images: int) -> torch.Tensor:
    B, S = input_ids.shape
    inc = torch.full((B, S), -1, dtype=torch.long)
    for b in range(B):
        count, seen_eod = -1, False
        for t in range(S):
            tok = int(input_ids[b, t])
            if tok == image_token_id:
                count += 1
                seen_eod = False
            inc[b, t] = count
            if seen_eod:
                inc[b, t] = -1
            if tok == eod_token_id:
                seen_eod = True
    inc[inc >= n_images] = -1
    neg = inc == -1
    inc = inc.masked_fill(neg, 0)
    mask = torch.nn.functional.one_hot(inc, num_classes=n_images)
    mask[neg, :] = 0
    return mask


def preprocess_images(u8, mean, std, rescale=1 / 255, valid_hw=None):
    """uint8 (n, H, W, 3) numpy -> (float32 (n, 3, H, W), mask (n, H, W) int64): hf:image_transforms.py rescale (:118-122:
    f32(f64(u8) * scale)) then normalize (:437: (x - f32 mean) / f32 std), channels first; pixels outside an image's valid
    (h, w) are zero with mask 0 (hf:idefics2 image processor's padding).  Pinned by tests/golden/g17_image_preprocess.npz."""
    import numpy as np
    x = (u8.astype(np.float64) * rescale).astype(np.float32)
    y = ((x - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)).transpose(0, 3, 1, 2)
    n, _, H, W = y.shape
    mask = np.ones((n, H, W), dtype=np.int64)
    if valid_hw is not None:
        yy, xx = np.arange(H)[None, :, None], np.arange(W)[None, None, :]
        mask = ((yy < valid_hw[:, 0, None, None]) & (xx < valid_hw[:, 1, None, None])).astype(np.int64)
        y = y * mask[:, None].astype(np.float32)
    return np.ascontiguousarray(y), mask
