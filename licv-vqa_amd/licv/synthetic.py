"""Seeded synthetic weights and VQA batches (no checkpoints / datasets are reachable offline).

Weights use the HF state-dict key names of ``IdeficsForVisionText2Text`` so a real
``idefics-9b`` checkpoint loads into the same engine unchanged
(names: hf:idefics/modeling_idefics.py:864-906, vision.py:75-90, perceiver.py:83-106).
Inputs follow SURVEY.md §8(d): right-padded rows, `<image>` wrapped by
`<fake_token_around_image>`, image_attention_mask by the incremental rule of
hf:idefics/processing_idefics.py:89-133.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .config import Idefics2Arch, IdeficsArch


def _randn(shape, std, gen, device, dtype, mean=0.0):
    t = torch.randn(shape, generator=gen, device=device, dtype=torch.float32)
    t = t * std + mean
    return t.to(dtype)


def synth_idefics_weights(arch: IdeficsArch, seed: int = 426, dtype=torch.bfloat16,
                          device="cpu", gate_std: float = 0.5) -> Dict[str, torch.Tensor]:
    """Random-init state dict.  Linear ~N(0,0.02); norm weights ~1+N(0,0.1); biases ~N(0,0.02);
    gated-x-attn alphas ~N(0,gate_std) so the tanh gates are open (HF init leaves them at 0,
    which would make the whole cross-attention path dead weight for parity purposes)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev if dev.type == "cuda" else "cpu")
    gen.manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, out_f, in_f, bias=False):
        sd[name + ".weight"] = _randn((out_f, in_f), 0.02, gen, dev, dtype)
        if bias:
            sd[name + ".bias"] = _randn((out_f,), 0.02, gen, dev, dtype)

    def norm(name, dim, bias=False):
        sd[name + ".weight"] = _randn((dim,), 0.1, gen, dev, dtype, mean=1.0)
        if bias:
            sd[name + ".bias"] = _randn((dim,), 0.02, gen, dev, dtype)

    a = arch
    H, I, hd = a.hidden_size, a.intermediate_size, a.head_dim
    sd["model.embed_tokens.weight"] = _randn((a.vocab_size, H), 0.02, gen, dev, dtype)
    if a.additional_vocab_size:
        sd["model.embed_tokens.additional_embedding.weight"] = _randn((a.additional_vocab_size, H), 0.02, gen, dev, dtype)
    # vision tower
    vp = "model.vision_model."
    sd[vp + "embeddings.class_embedding"] = _randn((a.v_embed,), 1.0, gen, dev, dtype)
    sd[vp + "embeddings.patch_embedding.weight"] = _randn((a.v_embed, 3, a.v_patch, a.v_patch), 0.02, gen, dev, dtype)
    sd[vp + "embeddings.position_embedding.weight"] = _randn((a.v_tokens, a.v_embed), 0.02, gen, dev, dtype)
    norm(vp + "pre_layrnorm", a.v_embed, True)
    for i in range(a.v_layers):
        p = f"{vp}encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(p + "self_attn." + n, a.v_embed, a.v_embed, True)
        norm(p + "layer_norm1", a.v_embed, True)
        lin(p + "mlp.fc1", a.v_inter, a.v_embed, True)
        lin(p + "mlp.fc2", a.v_embed, a.v_inter, True)
        norm(p + "layer_norm2", a.v_embed, True)
    norm(vp + "post_layernorm", a.v_embed, True)
    # perceiver
    if a.use_resampler:
        rp = "model.perceiver_resampler."
        sd[rp + "latents"] = _randn((a.r_latents, a.v_embed), 1.0, gen, dev, dtype)
        inner = a.r_heads * a.r_head_dim
        for i in range(a.r_depth):
            p = f"{rp}blocks.{i}.0."
            norm(p + "context_layer_norm", a.v_embed, True)
            norm(p + "latents_layer_norm", a.v_embed, True)
            if a.r_qk_norm:
                norm(p + "q_layer_norm", a.r_head_dim, True)
                norm(p + "k_layer_norm", a.r_head_dim, True)
            lin(p + "q_proj", inner, a.v_embed)
            lin(p + "k_proj", inner, a.v_embed)
            lin(p + "v_proj", inner, a.v_embed)
            lin(p + "output_proj", a.v_embed, inner)
            p = f"{rp}blocks.{i}.1."
            norm(p + "ln", a.v_embed, True)
            lin(p + "fc", 4 * a.v_embed, a.v_embed)
            lin(p + "c_proj", a.v_embed, 4 * a.v_embed)
        norm(rp + "layer_norm", a.v_embed, True)
    # decoder layers
    for i in range(a.num_layers):
        p = f"model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            lin(p + "self_attn." + n, H, H)
        lin(p + "mlp.gate_proj", I, H)
        lin(p + "mlp.down_proj", H, I)
        lin(p + "mlp.up_proj", I, H)
        norm(p + "input_layernorm", H)
        norm(p + "post_attention_layernorm", H)
    kv_in = a.v_embed
    for i in range(a.num_cross_layers):
        p = f"model.gated_cross_attn_layers.{i}."
        sd[p + "alpha_cross_attn"] = _randn((1,), gate_std, gen, dev, dtype)
        sd[p + "alpha_dense"] = _randn((1,), gate_std, gen, dev, dtype)
        lin(p + "cross_attn.q_proj", H, H)
        lin(p + "cross_attn.k_proj", H, kv_in)
        lin(p + "cross_attn.v_proj", H, kv_in)
        lin(p + "cross_attn.o_proj", H, H)
        if a.qk_layer_norms:
            norm(p + "cross_attn.q_layer_norm", hd)
            norm(p + "cross_attn.k_layer_norm", hd)
        lin(p + "mlp.gate_proj", I, H)
        lin(p + "mlp.down_proj", H, I)
        lin(p + "mlp.up_proj", I, H)
        norm(p + "input_layernorm", H)
        norm(p + "post_attention_layernorm", H)
    norm("model.norm", H)
    lin("lm_head", a.vocab_size, H)
    if a.additional_vocab_size:
        lin("lm_head.additional_fc", a.additional_vocab_size, H)
    return sd


_BRANCH_OUT = ("self_attn.o_proj.weight", "self_attn.out_proj.weight", "cross_attn.o_proj.weight", "mlp.down_proj.weight",
               "mlp.fc2.weight", ".c_proj.weight", ".output_proj.weight", "modality_projection.down_proj.weight")


def trained_like_(sd: Dict[str, torch.Tensor], depth: int) -> Dict[str, torch.Tensor]:
    """Scale every residual-branch OUTPUT projection by 1/sqrt(2*depth), in place (the GPT-2 / Llama-family initialisation
    that trained checkpoints keep the order of magnitude of): with all linears ~N(0, 0.02) a branch at width 4096 writes
    as much into the stream as the stream holds, and a 32-layer random model amplifies one bf16 ulp to percent-level
    logit differences — a property of the random init, not of any kernel.  Applied after generation, so the RNG stream
    (and every committed fixture) is unchanged."""
    f = 1.0 / math.sqrt(2.0 * max(depth, 1))
    for k, v in sd.items():
        if k.endswith(_BRANCH_OUT):
            v.mul_(f)
    return sd


def weights_checksum(sd: Dict[str, torch.Tensor]) -> float:
    """Order-independent fingerprint used by fixtures to detect RNG drift."""
    tot = 0.0
    for k in sorted(sd):
        tot += float(sd[k].double().abs().sum())
    return tot


def image_attention_mask_from_ids(input_ids: torch.Tensor, image_token_id: int, eos_token_id: int,
                                  n_images: int) -> torch.Tensor:
    """Vectorised form of the incremental rule (hf:idefics/processing_idefics.py:89-110 +
    incremental_to_binary_attention_mask :63-79): token t sees the most recent image; tokens before
    the first image, or after an EOS until the next image, see none.  Returns (B,S,n_images) int64."""
    is_img = input_ids == image_token_id
    count = is_img.long().cumsum(-1) - 1                              # index of most recent image
    is_eos = input_ids == eos_token_id
    # "seen_eod" at t: an EOS occurred strictly before t with no image token in (eos, t]
    pos = torch.arange(input_ids.shape[1], device=input_ids.device).expand_as(input_ids)
    neg = torch.full_like(pos, -1)
    last_img = torch.where(is_img, pos, neg).cummax(-1).values
    eos_before = torch.where(is_eos, pos, neg)
    eos_before = torch.cat([neg[:, :1], eos_before[:, :-1]], dim=1).cummax(-1).values
    seen = eos_before > last_img
    count = torch.where(seen, neg, count)
    mask = torch.zeros(*input_ids.shape, n_images, dtype=torch.long, device=input_ids.device)
    valid = (count >= 0) & (count < n_images)
    idx = count.clamp(min=0, max=max(n_images - 1, 0)).unsqueeze(-1)
    mask.scatter_(2, idx, valid.long().unsqueeze(-1))
    return mask


def synth_vqa_batch(arch: IdeficsArch, batch: int, seq_len: int, n_images: int, seed: int = 426,
                    min_len: Optional[int] = None, dtype=torch.bfloat16, device="cpu",
                    image_token_id: Optional[int] = None, fake_token_id: Optional[int] = None,
                    padding_side: str = "right") -> Dict[str, torch.Tensor]:
    """One synthetic VQA batch of the shape ``processor.prepare_input`` hands to the interface
    (ref:icv_src/icv_datamodule.py:80-124): input_ids, attention_mask, pixel_values,
    image_attention_mask.  Each image placeholder is `<fake><image><fake>` (ids default to the two
    additional-vocab slots)."""
    g = torch.Generator().manual_seed(seed)
    if image_token_id is None:
        image_token_id = arch.vocab_size + (1 if arch.additional_vocab_size > 1 else 0) if arch.additional_vocab_size else arch.vocab_size - 1
    if fake_token_id is None:
        fake_token_id = arch.vocab_size if arch.additional_vocab_size else arch.vocab_size - 2
    min_len = seq_len if min_len is None else min_len
    lo = 3
    hi = arch.vocab_size - (0 if arch.additional_vocab_size else 2)
    ids = torch.randint(lo, hi, (batch, seq_len), generator=g)
    lengths = torch.randint(min_len, seq_len + 1, (batch,), generator=g)
    ids[:, 0] = arch.bos_token_id
    # evenly spaced image slots, each needing 3 tokens; the first one right after BOS
    usable = int(lengths.min()) - 4
    assert usable >= 3 * n_images, "sequence too short for the requested number of images"
    stride = usable // n_images
    for k in range(n_images):
        p = 1 + k * stride
        ids[:, p] = fake_token_id
        ids[:, p + 1] = image_token_id
        ids[:, p + 2] = fake_token_id
    att = (torch.arange(seq_len).unsqueeze(0) < lengths.unsqueeze(1)).long()
    ids = torch.where(att.bool(), ids, torch.full_like(ids, arch.pad_token_id))
    if padding_side == "left":
        shift = seq_len - lengths
        idx = (torch.arange(seq_len).unsqueeze(0) - shift.unsqueeze(1)) % seq_len
        ids = ids.gather(1, idx)
        att = att.gather(1, idx)
    iam = image_attention_mask_from_ids(ids, image_token_id, -1, n_images)
    pix = torch.randn(batch, n_images, 3, arch.v_image, arch.v_image, generator=g)
    out = {
        "input_ids": ids.to(device),
        "attention_mask": att.to(device),
        "pixel_values": pix.to(dtype).to(device),
        "image_attention_mask": iam.to(device),
    }
    return out


def synth_icv(n_layers: int, hidden: int, seed: int = 426, alpha: float = 0.1, device="cpu"):
    """icv ~ N(0, 0.01) as ref:icv_src/icv_encoder/global_icv_encoder.py:30-31, constant alpha."""
    g = torch.Generator().manual_seed(seed + 7)
    icv = torch.randn(1, n_layers, hidden, generator=g) * 0.01
    al = torch.full((1, n_layers), float(alpha))
    return icv.to(device), al.to(device)


# ------------------------------------------------------------------------------------------ Idefics2
def synth_idefics2_weights(arch: Idefics2Arch, seed: int = 426, dtype=torch.bfloat16, device="cpu") -> Dict[str, torch.Tensor]:
    """Random-init state dict with the HF key names of ``Idefics2ForConditionalGeneration``
    (hf:idefics2/modeling_idefics2.py; text model = hf:mistral/modeling_mistral.py)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev if dev.type == "cuda" else "cpu")
    gen.manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    a = arch

    def lin(name, out_f, in_f, bias=False):
        sd[name + ".weight"] = _randn((out_f, in_f), 0.02, gen, dev, dtype)
        if bias:
            sd[name + ".bias"] = _randn((out_f,), 0.02, gen, dev, dtype)

    def norm(name, dim, bias=False):
        sd[name + ".weight"] = _randn((dim,), 0.1, gen, dev, dtype, mean=1.0)
        if bias:
            sd[name + ".bias"] = _randn((dim,), 0.02, gen, dev, dtype)

    vp = "model.vision_model."
    sd[vp + "embeddings.patch_embedding.weight"] = _randn((a.v_hidden, 3, a.v_patch, a.v_patch), 0.02, gen, dev, dtype)
    sd[vp + "embeddings.patch_embedding.bias"] = _randn((a.v_hidden,), 0.02, gen, dev, dtype)
    sd[vp + "embeddings.position_embedding.weight"] = _randn(((a.v_image // a.v_patch) ** 2, a.v_hidden), 0.02, gen, dev, dtype)
    for i in range(a.v_layers):
        p = f"{vp}encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(p + "self_attn." + n, a.v_hidden, a.v_hidden, True)
        norm(p + "layer_norm1", a.v_hidden, True)
        lin(p + "mlp.fc1", a.v_inter, a.v_hidden, True)
        lin(p + "mlp.fc2", a.v_hidden, a.v_inter, True)
        norm(p + "layer_norm2", a.v_hidden, True)
    norm(vp + "post_layernorm", a.v_hidden, True)
    cp = "model.connector."
    lin(cp + "modality_projection.gate_proj", a.intermediate_size, a.v_hidden)
    lin(cp + "modality_projection.up_proj", a.intermediate_size, a.v_hidden)
    lin(cp + "modality_projection.down_proj", a.hidden_size, a.intermediate_size)
    rp = cp + "perceiver_resampler."
    sd[rp + "latents"] = _randn((a.r_latents, a.hidden_size), 0.5, gen, dev, dtype, mean=1.0)
    for i in range(a.r_depth):
        p = f"{rp}layers.{i}."
        norm(p + "input_latents_norm", a.hidden_size)
        norm(p + "input_context_norm", a.hidden_size)
        lin(p + "self_attn.q_proj", a.r_heads * a.r_head_dim, a.hidden_size)
        lin(p + "self_attn.k_proj", a.r_kv_heads * a.r_head_dim, a.hidden_size)
        lin(p + "self_attn.v_proj", a.r_kv_heads * a.r_head_dim, a.hidden_size)
        lin(p + "self_attn.o_proj", a.hidden_size, a.r_heads * a.r_head_dim)
        norm(p + "post_attention_layernorm", a.hidden_size)
        lin(p + "mlp.gate_proj", 4 * a.hidden_size, a.hidden_size)
        lin(p + "mlp.up_proj", 4 * a.hidden_size, a.hidden_size)
        lin(p + "mlp.down_proj", a.hidden_size, 4 * a.hidden_size)
    norm(rp + "norm", a.hidden_size)
    tp = "model.text_model."
    sd[tp + "embed_tokens.weight"] = _randn((a.vocab_size, a.hidden_size), 0.02, gen, dev, dtype)
    hd = a.head_dim
    for i in range(a.num_layers):
        p = f"{tp}layers.{i}."
        lin(p + "self_attn.q_proj", a.num_heads * hd, a.hidden_size)
        lin(p + "self_attn.k_proj", a.num_kv_heads * hd, a.hidden_size)
        lin(p + "self_attn.v_proj", a.num_kv_heads * hd, a.hidden_size)
        lin(p + "self_attn.o_proj", a.hidden_size, a.num_heads * hd)
        lin(p + "mlp.gate_proj", a.intermediate_size, a.hidden_size)
        lin(p + "mlp.up_proj", a.intermediate_size, a.hidden_size)
        lin(p + "mlp.down_proj", a.hidden_size, a.intermediate_size)
        norm(p + "input_layernorm", a.hidden_size)
        norm(p + "post_attention_layernorm", a.hidden_size)
    norm(tp + "norm", a.hidden_size)
    lin("lm_head", a.vocab_size, a.hidden_size)
    return sd


def synth_vqa_batch_idefics2(arch: Idefics2Arch, batch: int, seq_len: int, n_images: int, img_h: int, img_w: int, seed: int = 426,
                             min_len: Optional[int] = None, dtype=torch.bfloat16, device="cpu", ragged: bool = True,
                             drop_last_image_of_row0: bool = False, padding_side: str = "right") -> Dict[str, torch.Tensor]:
    """input_ids with ``r_latents`` `<image>` tokens per image (hf:idefics2/processing_idefics2.py:129), right padded;
    pixel_values (B, N, 3, H, W) with per-image valid regions given by pixel_attention_mask (ragged NaViT images:
    the valid height/width are multiples of the patch size); optionally one all-zero padding image."""
    g = torch.Generator().manual_seed(seed)
    a = arch
    min_len = seq_len if min_len is None else min_len
    ids = torch.randint(3, a.image_token_id - 1, (batch, seq_len), generator=g)
    lengths = torch.randint(min_len, seq_len + 1, (batch,), generator=g)
    ids[:, 0] = a.bos_token_id
    need = n_images * (a.r_latents + 2)
    assert int(lengths.min()) - 2 >= need, "sequence too short for the requested images"
    n_img_row = [n_images] * batch
    if drop_last_image_of_row0 and n_images > 1:
        n_img_row[0] = n_images - 1
    for b in range(batch):
        p = 1
        for k in range(n_img_row[b]):
            ids[b, p + 1: p + 1 + a.r_latents] = a.image_token_id
            p += a.r_latents + 2
    att = (torch.arange(seq_len).unsqueeze(0) < lengths.unsqueeze(1)).long()
    ids = torch.where(att.bool(), ids, torch.full_like(ids, a.pad_token_id))
    if padding_side == "left":
        idx = (torch.arange(seq_len).unsqueeze(0) - (seq_len - lengths).unsqueeze(1)) % seq_len
        ids, att = ids.gather(1, idx), att.gather(1, idx)
    pix = torch.randn(batch, n_images, 3, img_h, img_w, generator=g)
    pam = torch.zeros(batch, n_images, img_h, img_w, dtype=torch.bool)
    P = a.v_patch
    for b in range(batch):
        for k in range(n_images):
            if k >= n_img_row[b]:
                pix[b, k] = 0.0                     # padding image: all zeros (removed by the model)
                continue
            hh = img_h if not ragged else P * int(torch.randint(max(1, img_h // P // 2), img_h // P + 1, (1,), generator=g))
            ww = img_w if not ragged else P * int(torch.randint(max(1, img_w // P // 2), img_w // P + 1, (1,), generator=g))
            pam[b, k, :hh, :ww] = True
            pix[b, k, :, hh:, :] = 0.0
            pix[b, k, :, :, ww:] = 0.0
    return {"input_ids": ids.to(device), "attention_mask": att.to(device), "pixel_values": pix.to(dtype).to(device),
            "pixel_attention_mask": pam.to(device)}
