"""Oracle (test infrastructure): Idefics forward with ICV hooks, restated functionally in torch-CPU.

Follows the installed transformers 5.15 Idefics code (``hf:`` =
transformers/models/idefics/) with ``attn_implementation="eager"``, op for op and dtype
for dtype, so that it can be pinned against HF-generated fixtures at tight tolerance in
both fp32 and bf16.  The model arithmetic is third-party to the reference
(transformers, pinned 4.38.2 in ref:requirements.txt:174); the hook arithmetic is the
reference's (``oracle.icv_ref.inject_renorm``).

Weights come as a flat state dict with HF key names.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from .icv_ref import inject_renorm


def _lin(x, sd, name):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def rms_norm(x, w, eps):
    """hf:idefics/modeling_idefics.py:342-350 — fp32 variance, x*rsqrt in promoted dtype, cast to
    the weight's half dtype *before* the weight multiply."""
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    x = x * torch.rsqrt(var + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        x = x.to(w.dtype)
    return w * x


def rotary_tables(head_dim: int, n_pos: int, base: float, dtype):
    """hf:idefics/modeling_idefics.py:357-393 — cat(freqs,freqs), cos/sin cast to the model dtype."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(torch.float) / head_dim))
    t = torch.arange(n_pos, dtype=torch.int64).type_as(inv_freq)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary(q, k, cos, sin, position_ids):
    """hf:idefics/modeling_idefics.py:403-428."""
    cos = cos[position_ids].unsqueeze(1)
    sin = sin[position_ids].unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def eager_attention(q, k, v, mask, scaling):
    """hf:idefics/modeling_idefics.py:450-470: matmul*scale (+mask) -> softmax fp32 -> cast -> matmul."""
    w = torch.matmul(q, k.transpose(-1, -2)) * scaling
    if mask is not None:
        w = w + mask
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v)
    return o.transpose(1, 2).contiguous()


def decoupled_embedding(ids, sd, vocab):
    """hf:idefics/modeling_idefics.py:230-267."""
    w = sd["model.embed_tokens.weight"]
    add = sd.get("model.embed_tokens.additional_embedding.weight")
    if add is None:
        return F.embedding(ids, w)
    hi = ids >= vocab
    base = F.embedding(torch.where(hi, torch.zeros_like(ids), ids), w)
    extra = F.embedding(torch.where(hi, ids - vocab, torch.zeros_like(ids)), add)
    return torch.where(hi.unsqueeze(-1), extra, base)


# ----------------------------------------------------------------------------- vision tower
def vision_tower(pixel_values, sd, arch):
    """hf:idefics/vision.py:142-166 (embeddings), :281-303 (layer), :341-381 (tower; returns
    last_hidden_state, i.e. *without* post_layernorm)."""
    p = "model.vision_model."
    wdt = sd[p + "embeddings.patch_embedding.weight"].dtype
    x = F.conv2d(pixel_values.to(wdt), sd[p + "embeddings.patch_embedding.weight"], None, stride=arch.v_patch)
    x = x.flatten(2).transpose(1, 2)
    cls = sd[p + "embeddings.class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1)
    x = x + sd[p + "embeddings.position_embedding.weight"].unsqueeze(0)
    x = F.layer_norm(x, (arch.v_embed,), sd[p + "pre_layrnorm.weight"], sd[p + "pre_layrnorm.bias"], arch.v_ln_eps)
    nh, hd = arch.v_heads, arch.v_head_dim
    for i in range(arch.v_layers):
        lp = f"{p}encoder.layers.{i}."
        res = x
        y = F.layer_norm(x, (arch.v_embed,), sd[lp + "layer_norm1.weight"], sd[lp + "layer_norm1.bias"], arch.v_ln_eps)
        B, T, _ = y.shape
        q = _lin(y, sd, lp + "self_attn.q_proj").view(B, T, nh, hd).transpose(1, 2)
        k = _lin(y, sd, lp + "self_attn.k_proj").view(B, T, nh, hd).transpose(1, 2)
        v = _lin(y, sd, lp + "self_attn.v_proj").view(B, T, nh, hd).transpose(1, 2)
        o = eager_attention(q, k, v, None, hd ** -0.5).reshape(B, T, -1).contiguous()
        x = res + _lin(o, sd, lp + "self_attn.out_proj")
        res = x
        y = F.layer_norm(x, (arch.v_embed,), sd[lp + "layer_norm2.weight"], sd[lp + "layer_norm2.bias"], arch.v_ln_eps)
        y = _lin(y, sd, lp + "mlp.fc1")
        y = F.gelu(y) if arch.v_act == "gelu" else F.gelu(y, approximate="tanh")
        x = res + _lin(y, sd, lp + "mlp.fc2")
    return x


# ----------------------------------------------------------------------------- perceiver
def perceiver(context, sd, arch):
    """hf:idefics/perceiver.py:93-103, :128-168, :171-187."""
    p = "model.perceiver_resampler."
    E, nh, hd = arch.v_embed, arch.r_heads, arch.r_head_dim
    lat = sd[p + "latents"].repeat(context.shape[0], 1, 1)
    for i in range(arch.r_depth):
        a = f"{p}blocks.{i}.0."
        ctx = F.layer_norm(context, (E,), sd[a + "context_layer_norm.weight"], sd[a + "context_layer_norm.bias"])
        l = F.layer_norm(lat, (E,), sd[a + "latents_layer_norm.weight"], sd[a + "latents_layer_norm.bias"])
        B = ctx.shape[0]
        q = _lin(l, sd, a + "q_proj")
        kv_in = torch.cat([ctx, l], dim=-2)
        k = _lin(kv_in, sd, a + "k_proj")
        v = _lin(kv_in, sd, a + "v_proj")
        q, k, v = [t.reshape(B, t.shape[1], nh, hd).transpose(1, 2) for t in (q, k, v)]
        if arch.r_qk_norm:
            q = F.layer_norm(q, (hd,), sd[a + "q_layer_norm.weight"], sd[a + "q_layer_norm.bias"])
            k = F.layer_norm(k, (hd,), sd[a + "k_layer_norm.weight"], sd[a + "k_layer_norm.bias"])
        scores = torch.einsum("... i d, ... j d -> ... i j", q * hd ** -0.5, k)
        scores = scores - scores.amax(dim=-1, keepdim=True)
        attn = scores.softmax(dim=-1)
        r = torch.einsum("... i j, ... j d -> ... i d", attn, v)
        lat = _lin(r.transpose(1, 2).flatten(-2), sd, a + "output_proj") + lat
        m = f"{p}blocks.{i}.1."
        y = F.layer_norm(lat, (E,), sd[m + "ln.weight"], sd[m + "ln.bias"])
        y = F.relu(_lin(y, sd, m + "fc"))
        lat = _lin(y, sd, m + "c_proj") + lat
    return F.layer_norm(lat, (E,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"])


# ----------------------------------------------------------------------------- masks
def build_masks(attention_mask, image_attention_mask, image_seq_len, dtype, q_len=None):
    """hf:idefics/modeling_idefics.py:972-1048.  Returns position_ids, additive causal mask
    (B,1,Sq,Sk), additive image mask (B,1,Sq,N_img*image_seq_len), cross_attention_gate (B,Sq)."""
    B, Sk = attention_mask.shape
    q_len = Sk if q_len is None else q_len
    position_ids = attention_mask.long().cumsum(-1) - 1
    position_ids = position_ids.masked_fill(attention_mask == 0, 1)[:, -q_len:]
    iam = image_attention_mask[..., None].expand(-1, -1, -1, image_seq_len)
    iam = iam.reshape(*image_attention_mask.shape[:2], -1)
    minv = torch.finfo(dtype).min
    img_mask = torch.where(iam[:, None, :, :].bool(), torch.full((), 0.0, dtype=dtype), minv)
    gate = (img_mask == 0.0).any(dim=-1).to(dtype).squeeze(1)
    kpos = torch.arange(Sk)
    qpos = torch.arange(Sk - q_len, Sk)
    allowed = (kpos[None, :] <= qpos[:, None])[None, None] & attention_mask.bool()[:, None, None, :]
    causal = torch.where(allowed, torch.full((), 0.0, dtype=dtype), minv)
    return position_ids, causal, img_mask, gate


# ----------------------------------------------------------------------------- LM blocks
def _mlp(x, sd, p):
    return _lin(F.silu(_lin(x, sd, p + "gate_proj")) * _lin(x, sd, p + "up_proj"), sd, p + "down_proj")


def decoder_layer(h, sd, i, arch, causal, position_ids, cos, sin, kv_cache=None):
    """hf:idefics/modeling_idefics.py:645-675 (+ attention :561-620)."""
    p = f"model.layers.{i}."
    nh, hd = arch.num_heads, arch.head_dim
    res = h
    x = rms_norm(h, sd[p + "input_layernorm.weight"], arch.rms_eps)
    B, T, _ = x.shape
    q = _lin(x, sd, p + "self_attn.q_proj").view(B, T, nh, hd).transpose(1, 2)
    k = _lin(x, sd, p + "self_attn.k_proj").view(B, T, nh, hd).transpose(1, 2)
    v = _lin(x, sd, p + "self_attn.v_proj").view(B, T, nh, hd).transpose(1, 2)
    q, k = apply_rotary(q, k, cos.to(v.dtype), sin.to(v.dtype), position_ids)
    if kv_cache is not None:
        if kv_cache[i] is not None:
            k = torch.cat([kv_cache[i][0], k], dim=2)
            v = torch.cat([kv_cache[i][1], v], dim=2)
        kv_cache[i] = (k, v)
    o = eager_attention(q, k, v, causal, hd ** -0.5).reshape(B, T, -1).contiguous()
    h = res + _lin(o, sd, p + "self_attn.o_proj")
    res = h
    x = rms_norm(h, sd[p + "post_attention_layernorm.weight"], arch.rms_eps)
    return res + _mlp(x, sd, p + "mlp.")


def gated_xattn_layer(h, sd, j, arch, image_states, img_mask, gate):
    """hf:idefics/modeling_idefics.py:746-802."""
    p = f"model.gated_cross_attn_layers.{j}."
    nh, hd = arch.num_heads, arch.head_dim
    res = h
    x = rms_norm(h, sd[p + "input_layernorm.weight"], arch.rms_eps)
    B, T, _ = x.shape
    Tk = image_states.shape[1]
    q = _lin(x, sd, p + "cross_attn.q_proj").view(B, T, nh, hd).transpose(1, 2)
    k = _lin(image_states, sd, p + "cross_attn.k_proj").view(B, Tk, nh, hd).transpose(1, 2)
    v = _lin(image_states, sd, p + "cross_attn.v_proj").view(B, Tk, nh, hd).transpose(1, 2)
    if arch.qk_layer_norms:
        q = rms_norm(q, sd[p + "cross_attn.q_layer_norm.weight"], arch.rms_eps)
        k = rms_norm(k, sd[p + "cross_attn.k_layer_norm.weight"], arch.rms_eps)
    o = eager_attention(q, k, v, img_mask, hd ** -0.5).reshape(B, T, -1).contiguous()
    x = _lin(o, sd, p + "cross_attn.o_proj")
    x = x.masked_fill((gate == 0)[:, :, None], 0.0)
    h = res + torch.tanh(sd[p + "alpha_cross_attn"]) * x
    res = h
    x = rms_norm(h, sd[p + "post_attention_layernorm.weight"], arch.rms_eps)
    return res + torch.tanh(sd[p + "alpha_dense"]) * _mlp(x, sd, p + "mlp.")


def lm_head(x, sd):
    """hf:idefics/modeling_idefics.py:318-325."""
    out = F.linear(x, sd["lm_head.weight"])
    if "lm_head.additional_fc.weight" in sd:
        out = torch.cat((out, F.linear(x, sd["lm_head.additional_fc.weight"])), -1)
    return out


def image_states_from_pixels(pixel_values, sd, arch):
    dtype = sd["model.embed_tokens.weight"].dtype
    B, N = pixel_values.shape[:2]
    pv = pixel_values.to(dtype).contiguous().view(B * N, *pixel_values.shape[2:])
    x = vision_tower(pv, sd, arch)
    if arch.use_resampler:
        x = perceiver(x, sd, arch)
    return x.view(B, N * x.shape[1], x.shape[2])


def forward(sd: Dict[str, torch.Tensor], arch, input_ids, attention_mask, pixel_values=None,
            image_attention_mask=None, icv: Optional[torch.Tensor] = None,
            hook_layers: Optional[Sequence[int]] = None, capture: Optional[dict] = None,
            image_states: Optional[torch.Tensor] = None, kv_cache: Optional[list] = None,
            n_layers: Optional[int] = None):
    """IdeficsForVisionText2Text.forward (hf:idefics/modeling_idefics.py:934-1084, :1165-1182) with the
    reference hook applied to the *output of decoder layer l* for l in ``hook_layers``
    (ref:config/lmm/idefics-9B.yaml:7, ref:icv_src/icv_model/icv_intervention.py:61-86).
    ``icv`` is (1, len(hook_layers), H), already alpha-scaled.  ``capture`` (dict) receives
    'raw' / 'edited' lists of per-layer outputs.  ``kv_cache`` (list of len L) enables decode steps:
    pass only the new ids; attention_mask always spans past+new."""
    dtype = sd["model.embed_tokens.weight"].dtype
    L = arch.num_layers if n_layers is None else n_layers
    h = decoupled_embedding(input_ids, sd, arch.vocab_size)
    q_len = input_ids.shape[1]
    if image_states is None:
        image_states = image_states_from_pixels(pixel_values, sd, arch)
    position_ids, causal, img_mask, gate = build_masks(
        attention_mask, image_attention_mask, arch.image_seq_len, image_states.dtype, q_len)
    causal = causal.to(dtype)
    cos, sin = rotary_tables(arch.head_dim, max(arch.max_positions, attention_mask.shape[1]), arch.rope_base, dtype)
    idx_of = {int(l): i for i, l in enumerate(hook_layers)} if (icv is not None and hook_layers is not None) else {}
    for l in range(L):
        if l % arch.cross_layer_interval == 0:
            h = gated_xattn_layer(h, sd, l // arch.cross_layer_interval, arch, image_states, img_mask, gate.to(dtype))
        h = decoder_layer(h, sd, l, arch, causal, position_ids, cos, sin, kv_cache)
        if capture is not None:
            capture.setdefault("raw", []).append(h)
        if l in idx_of:
            h = inject_renorm(h, icv[:, idx_of[l]].unsqueeze(1))
        if capture is not None:
            capture.setdefault("edited", []).append(h)
    h = rms_norm(h, sd["model.norm.weight"], arch.rms_eps)
    if capture is not None:
        capture["final_norm"] = h
        capture["image_states"] = image_states
    return lm_head(h, sd)
