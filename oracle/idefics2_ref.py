"""Oracle (test infrastructure): Idefics2 forward with the ICV hook on every text layer's ``.mlp`` output.

Restates transformers 5.15 ``Idefics2ForConditionalGeneration`` (``hf:`` = transformers/models/idefics2/,
text model hf:mistral/modeling_mistral.py) with eager attention, functionally, from a flat HF-named state dict.
Call it inside ``torch.autocast("cpu", dtype=torch.bfloat16)`` for the bf16 path: with the fp32 ICV promoting the
residual stream the HF model only runs under autocast, and the same context makes these
functions round exactly where HF does.  Hook site: ref:config/lmm/idefics2-8B-base.yaml:8
(``model.model.text_model.layers.<N>.mlp``, i.e. the MLP branch BEFORE the residual add).
Pinned by tests/golden/g4_idefics2_*.npz (made by the reference's wrapper driving HF).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from .icv_ref import inject_renorm
from .idefics_ref import _lin, eager_attention, rotate_half


def rms_norm(x, w, eps):
    """hf:mistral/modeling_mistral.py:182-199 / hf:idefics2/modeling_idefics2.py Idefics2RMSNorm: fp32 statistics, cast
    back to the INPUT dtype, then weight * x."""
    dt = x.dtype
    x = x.to(torch.float32)
    var = x.pow(2).mean(-1, keepdim=True)
    x = x * torch.rsqrt(var + eps)
    return w * x.to(dt)


def repeat_kv(x, n_rep):
    if n_rep == 1:
        return x
    b, h, s, d = x.shape
    return x[:, :, None].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def _additive(mask_bool, dtype):
    """(B, Sk) bool key-valid -> (B,1,1,Sk) additive 0 / finfo.min, or None when everything is valid."""
    if bool(mask_bool.all()):
        return None
    return torch.where(mask_bool[:, None, None, :], torch.zeros((), dtype=dtype), torch.finfo(dtype).min)


# ----------------------------------------------------------------------------- vision (hf:idefics2/modeling_idefics2.py:100-173, :341-362, :430-500)
def patch_mask_from_pixels(pixel_attention_mask, patch):
    sub = pixel_attention_mask.unfold(1, patch, patch).unfold(2, patch, patch)
    return (sub.sum(dim=(-1, -2)) == patch * patch).bool()


def navit_position_ids(patch_mask, n_side):
    """hf:idefics2/modeling_idefics2.py:136-170 (fractional coordinates bucketised into the n_side x n_side grid)."""
    n, gh, gw = patch_mask.shape
    boundaries = torch.arange(1 / n_side, 1.0, 1 / n_side)
    nb_h = patch_mask[:, :, 0].sum(dim=1)
    nb_w = patch_mask[:, 0, :].sum(dim=1)
    fh = torch.arange(gh, dtype=torch.float32)[None, :] * (1.0 / nb_h)[:, None]
    fw = torch.arange(gw, dtype=torch.float32)[None, :] * (1.0 / nb_w)[:, None]
    fh = torch.clamp(fh, max=1.0 - 1e-6)
    fw = torch.clamp(fw, max=1.0 - 1e-6)
    return fh, fw, boundaries


# ----------------------------------------------------------------------------- fp8 projections (the build's configs[4] mode)
def fp8_linear(x, w, bias=None):
    """CPU restatement of the build's fp8 text-stack projection (BASELINE configs[4]; the reference has no fp8 mode, so this
    pins the ARITHMETIC the HIP path claims, csrc/rowwise.hip quantize_rows_fp8_k / emit_row_fp8 + csrc/gemm.hip gemm_fp8_flow64_k,
    gemm_fp8_pingpong_k):
    per-row dynamic activation scale amax/448 and per-output-channel weight scale amax/448, both operands rounded to OCP
    e4m3 (round to nearest even), exact products accumulated in fp32, scales applied to the accumulator, one rounding to the
    activation dtype."""
    if torch.is_autocast_enabled("cpu"):                 # autocast hands a linear bf16 operands, and returns bf16
        x, w = x.to(torch.bfloat16), w.to(torch.bfloat16)
    dt = x.dtype
    with torch.autocast("cpu", enabled=False):
        xf, wf = x.float(), w.float()
        sx = xf.abs().amax(-1, keepdim=True).clamp_min(1e-12) / 448.0
        sw = wf.abs().amax(-1, keepdim=True).clamp_min(1e-12) / 448.0
        xq = (xf / sx).to(torch.float8_e4m3fn).float()
        wq = (wf / sw).to(torch.float8_e4m3fn).float()
        y = (xq @ wq.t()) * sx * sw.t()
        if bias is not None:                             # the build adds the bias to the scaled fp32 accumulator: ONE rounding
            y = y + bias.float()
    return y.to(dt)



def vision_tower(pixel_values, patch_mask, sd, arch, fp8: bool = False):
    """fp8: the four projections of every layer through ``fp8_linear`` (the build's fp8_vision mode; not a reference feature)."""
    lin = (lambda x_, sd_, name: fp8_linear(x_, sd_[name + ".weight"], sd_.get(name + ".bias"))) if fp8 else _lin
    p = "model.vision_model."
    n = pixel_values.shape[0]
    x = F.conv2d(pixel_values, sd[p + "embeddings.patch_embedding.weight"], sd[p + "embeddings.patch_embedding.bias"], stride=arch.v_patch)
    x = x.flatten(2).transpose(1, 2)
    n_side = arch.v_image // arch.v_patch
    fh, fw, boundaries = navit_position_ids(patch_mask, n_side)
    fh, fw = fh.to(pixel_values.dtype), fw.to(pixel_values.dtype)
    bh = torch.bucketize(fh, boundaries, right=True)
    bw = torch.bucketize(fw, boundaries, right=True)
    pos = (bh[:, :, None] * n_side + bw[:, None, :]).reshape(n, -1)
    flat = patch_mask.view(n, -1)
    position_ids = torch.zeros_like(pos)
    position_ids[flat] = pos[flat]
    x = x + F.embedding(position_ids, sd[p + "embeddings.position_embedding.weight"])
    mask = _additive(flat, x.dtype)
    nh, hd = arch.v_heads, arch.v_head_dim
    for i in range(arch.v_layers):
        lp = f"{p}encoder.layers.{i}."
        res = x
        y = F.layer_norm(x, (arch.v_hidden,), sd[lp + "layer_norm1.weight"], sd[lp + "layer_norm1.bias"], arch.v_ln_eps)
        B, T, _ = y.shape
        q = lin(y, sd, lp + "self_attn.q_proj").view(B, T, nh, hd).transpose(1, 2)
        k = lin(y, sd, lp + "self_attn.k_proj").view(B, T, nh, hd).transpose(1, 2)
        v = lin(y, sd, lp + "self_attn.v_proj").view(B, T, nh, hd).transpose(1, 2)
        m = None if mask is None else mask.to(q.dtype)
        o = eager_attention(q, k, v, m, hd ** -0.5).reshape(B, T, -1).contiguous()
        x = res + lin(o, sd, lp + "self_attn.out_proj")
        res = x
        y = F.layer_norm(x, (arch.v_hidden,), sd[lp + "layer_norm2.weight"], sd[lp + "layer_norm2.bias"], arch.v_ln_eps)
        y = F.gelu(lin(y, sd, lp + "mlp.fc1"), approximate="tanh")
        x = res + lin(y, sd, lp + "mlp.fc2")
    return F.layer_norm(x, (arch.v_hidden,), sd[p + "post_layernorm.weight"], sd[p + "post_layernorm.bias"], arch.v_ln_eps)


def _mlp(x, sd, p):
    return _lin(F.silu(_lin(x, sd, p + "gate_proj")) * _lin(x, sd, p + "up_proj"), sd, p + "down_proj")


def connector(x, patch_valid, sd, arch):
    """hf:idefics2/modeling_idefics2.py:757-760 (modality projection) + :708-743 (perceiver resampler)."""
    cp = "model.connector."
    ctx = _mlp(x, sd, cp + "modality_projection.")
    rp = cp + "perceiver_resampler."
    n = ctx.shape[0]
    lat = sd[rp + "latents"].unsqueeze(0).expand(n, -1, -1)
    valid = torch.cat([patch_valid, torch.ones((n, arch.r_latents), dtype=patch_valid.dtype)], dim=-1).bool()
    mask = _additive(valid, lat.dtype)
    nh, nkv, hd = arch.r_heads, arch.r_kv_heads, arch.r_head_dim
    for i in range(arch.r_depth):
        lp = f"{rp}layers.{i}."
        res = lat
        l = rms_norm(lat, sd[lp + "input_latents_norm.weight"], arch.rms_eps)
        c = rms_norm(ctx, sd[lp + "input_context_norm.weight"], arch.rms_eps)
        hs = torch.concat([c, l], dim=-2)
        B, Lq, _ = l.shape
        q = _lin(l, sd, lp + "self_attn.q_proj").view(B, Lq, nh, hd).transpose(1, 2)
        k = _lin(hs, sd, lp + "self_attn.k_proj").view(B, hs.shape[1], nkv, hd).transpose(1, 2)
        v = _lin(hs, sd, lp + "self_attn.v_proj").view(B, hs.shape[1], nkv, hd).transpose(1, 2)
        m = None if mask is None else mask.to(q.dtype)
        o = eager_attention(q, repeat_kv(k, nh // nkv), repeat_kv(v, nh // nkv), m, hd ** -0.5).reshape(B, Lq, nh * hd)
        lat = res + _lin(o, sd, lp + "self_attn.o_proj")
        res = lat
        lat = res + _mlp(rms_norm(lat, sd[lp + "post_attention_layernorm.weight"], arch.rms_eps), sd, lp + "mlp.")
    return rms_norm(lat, sd[rp + "norm.weight"], arch.rms_eps)


def image_features(pixel_values, pixel_attention_mask, sd, arch, fp8_vision: bool = False):
    """hf:idefics2/modeling_idefics2.py:817-862: drop all-zero padding images, patch mask, tower, connector.
    Returns (n_real_images * r_latents, H)."""
    dtype = sd["model.text_model.embed_tokens.weight"].dtype
    B, N = pixel_values.shape[:2]
    pv = pixel_values.to(dtype).view(B * N, *pixel_values.shape[2:])
    real = (pv == 0.0).sum(dim=(-1, -2, -3)) != pv.shape[1:].numel()
    pv = pv[real].contiguous()
    pam = pixel_attention_mask.view(B * N, *pixel_attention_mask.shape[2:])[real].contiguous()
    pmask = patch_mask_from_pixels(pam, arch.v_patch)
    x = vision_tower(pv, pmask, sd, arch, fp8=fp8_vision)
    feats = connector(x, pmask.view(pv.shape[0], -1), sd, arch)
    return feats.reshape(-1, feats.shape[-1])


# ----------------------------------------------------------------------------- Mistral text model
def forward(sd: Dict[str, torch.Tensor], arch, input_ids, attention_mask, pixel_values=None, pixel_attention_mask=None,
            icv: Optional[torch.Tensor] = None, hook_layers: Optional[Sequence[int]] = None, capture: Optional[dict] = None,
            image_hidden_states: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None,
            fp8_text: bool = False, scatter_len: Optional[int] = None, fp8_vision: bool = False):
    """logits (B, S, V).  icv (1, n_hooked, H) fp32, already alpha-scaled; the hook edits the MLP output of text layer l.
    fp8_text: the four projections of every text layer run through ``fp8_linear`` (the build's configs[4] mode).
    scatter_len: only `<image>` tokens at positions < scatter_len receive image features (a cache-less decode re-runs the whole
    prefix; HF merges image features in the prefill only, so an `<image>` id that was GENERATED keeps its table embedding)."""
    tp = "model.text_model."
    B, S = input_ids.shape
    h = F.embedding(input_ids, sd[tp + "embed_tokens.weight"])
    if image_hidden_states is None and pixel_values is not None:
        image_hidden_states = image_features(pixel_values, pixel_attention_mask, sd, arch, fp8_vision=fp8_vision)
    if image_hidden_states is not None:
        special = input_ids == arch.image_token_id
        if scatter_len is not None:
            special = special & (torch.arange(S)[None, :] < scatter_len)
        h = h.masked_scatter(special.unsqueeze(-1), image_hidden_states.to(h.dtype))
    dtype = h.dtype
    nh, nkv, hd = arch.num_heads, arch.num_kv_heads, arch.head_dim
    # rotary from position_ids = arange(S) (hf:mistral/modeling_mistral.py MistralModel.forward), fp32 maths, cast to the model dtype
    inv = 1.0 / (arch.rope_base ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    if position_ids is None:                              # plain forward; generate passes HF's mask-derived ids (B, S)
        position_ids = torch.arange(S)[None, :]
    freqs = position_ids[..., None].to(torch.float) * inv
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos()[:, None].to(dtype), emb.sin()[:, None].to(dtype)
    minv = torch.finfo(dtype).min
    allowed = torch.tril(torch.ones(S, S, dtype=torch.bool))[None, None] & attention_mask.bool()[:, None, None, :]
    causal = torch.where(allowed, torch.zeros((), dtype=dtype), minv)
    idx_of = {int(l): i for i, l in enumerate(hook_layers)} if (icv is not None and hook_layers is not None) else {}
    lin = (lambda x_, sd_, name: fp8_linear(x_, sd_[name + ".weight"])) if fp8_text else _lin
    mlp = (lambda x_, sd_, p_: lin(F.silu(lin(x_, sd_, p_ + "gate_proj")) * lin(x_, sd_, p_ + "up_proj"), sd_, p_ + "down_proj")) if fp8_text else _mlp
    for l in range(arch.num_layers):
        lp = f"{tp}layers.{l}."
        res = h
        x = rms_norm(h, sd[lp + "input_layernorm.weight"], arch.rms_eps)
        q = lin(x, sd, lp + "self_attn.q_proj").view(B, S, nh, hd).transpose(1, 2)
        k = lin(x, sd, lp + "self_attn.k_proj").view(B, S, nkv, hd).transpose(1, 2)
        v = lin(x, sd, lp + "self_attn.v_proj").view(B, S, nkv, hd).transpose(1, 2)
        c, s_ = cos.to(q.dtype), sin.to(q.dtype)
        q, k = (q * c) + (rotate_half(q) * s_), (k * c) + (rotate_half(k) * s_)
        o = eager_attention(q, repeat_kv(k, nh // nkv), repeat_kv(v, nh // nkv), causal.to(q.dtype), hd ** -0.5).reshape(B, S, -1).contiguous()
        h = res + lin(o, sd, lp + "self_attn.o_proj")
        res = h
        m = mlp(rms_norm(h, sd[lp + "post_attention_layernorm.weight"], arch.rms_eps), sd, lp + "mlp.")
        if capture is not None:
            capture.setdefault("mlp_raw", []).append(m)
        if l in idx_of:
            m = inject_renorm(m, icv[:, idx_of[l]].unsqueeze(1))
        h = res + m
        if capture is not None:
            capture.setdefault("layer_out", []).append(h)
    h = rms_norm(h, sd[tp + "norm.weight"], arch.rms_eps)
    if capture is not None:
        capture["image_hidden_states"] = image_hidden_states
    return F.linear(h, sd["lm_head.weight"])
