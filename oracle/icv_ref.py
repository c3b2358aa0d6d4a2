"""Oracle (test infrastructure): the ICV arithmetic the reference owns, restated in plain torch-CPU.

Every function cites the reference lines it follows (``ref:`` = /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch


# --------------------------------------------------------------------------------------
# encoder  (ref:icv_src/icv_encoder/global_icv_encoder.py:6-43)
# --------------------------------------------------------------------------------------
def encoder_init(hidden: int, layers: int, alpha_init_value: float = 0.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """alpha (1,L)=full(alpha_init_value) is created first, then icv (1,L,H) ~ N(0, 0.01) drawn
    from the global torch RNG (ref :26-31) — so under the same ``torch.manual_seed`` the values
    are bit-identical to the reference module's."""
    alpha = torch.full(size=(1, layers), fill_value=float(alpha_init_value))
    icv = torch.empty(1, layers, hidden)
    torch.nn.init.normal_(icv, mean=0.0, std=0.01)
    return icv, alpha


def encoder_alpha(alpha: torch.Tensor, use_sigmoid: bool) -> torch.Tensor:
    """ref :40-43."""
    return torch.sigmoid(alpha) if use_sigmoid else alpha


def scale_icv(alpha: torch.Tensor, icv: torch.Tensor) -> torch.Tensor:
    """icv_eff = alpha.unsqueeze(-1) * in_context_vector  (ref:icv_src/icv_module.py:89-92,
    ref:inference.py:309-311)."""
    return alpha.unsqueeze(dim=-1) * icv


# --------------------------------------------------------------------------------------
# layer bookkeeping (ref:icv_src/icv_model/icv_intervention.py:22-42)
# --------------------------------------------------------------------------------------
def prepare_layers(intervention_layer, total_layers: int) -> List[int]:
    if intervention_layer == -1:
        return list(range(total_layers))
    return [intervention_layer] if isinstance(intervention_layer, int) else list(intervention_layer)


def layer_names(layers: Sequence[int], layer_format: str) -> List[str]:
    return [layer_format.replace("<LAYER_NUM>", str(l)) for l in layers]


def layer_to_icv_index(layers: Sequence[int]) -> Dict[int, int]:
    return {int(l): int(i) for i, l in enumerate(layers)}


# --------------------------------------------------------------------------------------
# the hook  (ref:icv_src/icv_model/icv_intervention.py:61-86)
# --------------------------------------------------------------------------------------
def inject_renorm(h: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """h' = (h+v)/||h+v|| * ||h||, norms over the last dim; `shift` is (1,1,H) or (H,).  Type
    promotion is torch's own (bf16 h + fp32 v -> fp32), exactly as the reference relies on."""
    shift = shift.reshape(1, 1, -1)
    shifted = h + shift
    return shifted / shifted.norm(dim=-1, keepdim=True) * h.norm(dim=-1, keepdim=True)


def inject_renorm_bwd(h: torch.Tensor, shift: torch.Tensor, grad_out: torch.Tensor):
    """Analytic backward of :func:`inject_renorm` in fp64 (checked against autograd in the tests).
    Returns (grad_h, grad_shift[H])."""
    h64, v64, g64 = h.double(), shift.reshape(1, 1, -1).double(), grad_out.double()
    s = h64 + v64
    ns = s.norm(dim=-1, keepdim=True)
    # torch's .norm() returns the input dtype: for a bf16 stream the VALUE of ||h|| is bf16-rounded
    nh = h.norm(dim=-1, keepdim=True).double() if h.dtype == torch.bfloat16 else h64.norm(dim=-1, keepdim=True)
    u = s / ns
    gu = (g64 * u).sum(-1, keepdim=True)
    gs = (nh / ns) * (g64 - u * gu)                 # through s/||s||
    gh = gs + (gu / nh) * h64                       # + through ||h||
    gv = gs.reshape(-1, gs.shape[-1]).sum(0)
    return gh, gv


# --------------------------------------------------------------------------------------
# masks and losses  (ref:icv_src/icv_module.py:121-148)
# --------------------------------------------------------------------------------------
def get_mask(input_ids: torch.Tensor, mask_length: torch.Tensor, pad_token_id: int) -> torch.Tensor:
    """mask[b,t] = (t >= length[b]) & (ids[b,t] != pad)   (ref :136-148)."""
    bs, seq_len = input_ids.shape
    idx = torch.arange(seq_len, device=input_ids.device).unsqueeze(0).expand(bs, -1)
    mask = idx >= mask_length.unsqueeze(dim=1)
    mask = mask & (input_ids != pad_token_id)
    return mask


def kl_divergence(stu_logits: torch.Tensor, tea_logits: torch.Tensor, temperature: float = 1.0,
                  eps: float = 1e-6) -> torch.Tensor:
    """mean_rows sum_v p (log(p+eps) - log(q+eps)) * T^2, softmax in the input dtype, eps inside the
    log  (ref :121-134).  Inputs are (rows, V); not modified (the reference's in-place `/=` acts on
    copies made by boolean indexing)."""
    stu = stu_logits / temperature
    tea = tea_logits / temperature
    p = tea.softmax(dim=1)
    q = stu.softmax(dim=1)
    return (p * ((p + eps).log() - (q + eps).log())).sum(dim=1).mean() * temperature ** 2


def ce_masked(logits: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """Shift-by-one CE with labels=input_ids, positions masked by attention_mask[:,1:], mean over
    kept positions in fp32 — the behaviour of the reference's pinned transformers 4.38.2 Idefics
    forward (equals HF 5.x ``ForCausalLMLoss`` with pad labels set to -100)."""
    shift_logits = logits[:, :-1, :].float()
    shift_labels = input_ids[:, 1:]
    keep = attention_mask[:, 1:] != 0
    return torch.nn.functional.cross_entropy(shift_logits[keep], shift_labels[keep], reduction="mean")


# --------------------------------------------------------------------------------------
# optimiser  (ref:icv_src/icv_module.py:171-209; torch.optim.AdamW; transformers
# get_cosine_schedule_with_warmup)
# --------------------------------------------------------------------------------------
def cosine_warmup_lambda(step: int, warmup: float, total: float, num_cycles: float = 0.5) -> float:
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
               beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 1e-3):
    """One decoupled-weight-decay Adam step (torch.optim.AdamW single-tensor semantics); `step` is
    1-based.  Returns new (p, m, v)."""
    p = p * (1.0 - lr * weight_decay)
    m = m * beta1 + (1.0 - beta1) * g
    v = v * beta2 + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def clip_grad_norm(grads: Sequence[torch.Tensor], max_norm: float = 1.0) -> Tuple[List[torch.Tensor], float]:
    """Global L2 clip as torch.nn.utils.clip_grad_norm_ (trainer/*.yaml: gradient_clip_val 1.0)."""
    total = math.sqrt(sum(float(g.double().pow(2).sum()) for g in grads))
    coef = min(1.0, max_norm / (total + 1e-6))
    return [g * coef for g in grads], total
