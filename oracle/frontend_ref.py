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
