"""Algorithmic work of the hooked Idefics forward, recomputed from the live config (SURVEY.md §8d formulas).

All figures are per QUESTION (one batch row: S text tokens, N_img images)."""
from __future__ import annotations

from .config import IdeficsArch

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0           # HBM3E spec peak (same table); ~6300 GB/s is the measured float4-copy ceiling


def flops_per_question(a: IdeficsArch, S: int, n_img: int) -> dict:
    H, I, V = a.hidden_size, a.intermediate_size, a.total_vocab
    E, T = a.v_embed, a.v_tokens
    Lq = a.image_seq_len
    p_dec = 4 * H * H + 3 * H * I
    p_x_text = 2 * H * H + 3 * H * I
    p_x_img = 2 * E * H
    lm = 2 * S * (a.num_layers * p_dec + a.num_cross_layers * p_x_text + H * V)
    xkv = 2 * (n_img * Lq) * a.num_cross_layers * p_x_img
    self_attn = a.num_layers * 2 * S * S * H                       # causal: half of 4*S*S*H
    x_attn = a.num_cross_layers * 4 * S * (n_img * Lq) * H
    p_vit = a.v_layers * (4 * E * E + 2 * E * a.v_inter)
    vit = n_img * (T * 2 * (p_vit + 3 * a.v_patch * a.v_patch * E) + a.v_layers * 4 * T * T * E)
    perc = 0
    if a.use_resampler:
        inner = a.r_heads * a.r_head_dim
        perc = n_img * a.r_depth * (2 * Lq * E * inner * 2 + 2 * (T + Lq) * E * inner * 2 + 4 * Lq * (T + Lq) * inner
                                    + 2 * Lq * 2 * E * 4 * E)
    parts = dict(lm_dense=lm, xattn_kv=xkv, self_attn=self_attn, cross_attn=x_attn, vision=vit, perceiver=perc)
    parts["total"] = sum(parts.values())
    return parts


def inject_bytes_per_question(a: IdeficsArch, S: int, n_hooked: int, first_stream_bf16: bool = True) -> int:
    """Algorithmic bytes of the hook = 1 read + 1 write of h per hooked layer: the first hooked
    layer reads a bf16 stream, all later ones fp32; outputs are fp32."""
    H = a.hidden_size
    if n_hooked == 0:
        return 0
    first = S * H * ((2 if first_stream_bf16 else 4) + 4)
    return first + (n_hooked - 1) * S * H * 8


def weight_bytes(a: IdeficsArch) -> dict:
    """bf16 bytes of the weight matrices one forward streams, per tower (the bound of the weight-streaming shapes of SURVEY.md
    §8d: the 32-token student and the decode steps of hooked generate read every weight once per pass and reuse it over few rows)."""
    H, I, V = a.hidden_size, a.intermediate_size, a.total_vocab
    E = a.v_embed
    lm = a.num_layers * (4 * H * H + 3 * H * I) + a.num_cross_layers * (2 * H * H + 3 * H * I + 2 * E * H) + H * V
    vit = a.v_layers * (4 * E * E + 2 * E * a.v_inter) + 3 * a.v_patch * a.v_patch * E
    perc = 0
    if a.use_resampler:
        inner = a.r_heads * a.r_head_dim
        perc = a.r_depth * (4 * E * inner + 2 * E * 4 * E)
    return dict(language=2 * lm, vision=2 * vit, perceiver=2 * perc, total=2 * (lm + vit + perc))


def cross_kv_weight_bytes(a: IdeficsArch) -> int:
    """The cross-attention K|V projections (image side): computed at the prefill only, the decode steps reuse their output."""
    return 2 * a.num_cross_layers * 2 * a.v_embed * a.hidden_size
