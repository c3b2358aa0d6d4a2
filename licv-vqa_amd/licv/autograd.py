"""torch.autograd bridges over the native student pass, so the reference's call shape works unchanged:

    loss_dict, _ = VQAICVModule.forward(...);  loss_dict["loss"].backward()      (ref:icv_src/icv_module.py:97-118,160-169)

fills ``icv_encoder.icv.grad`` / ``icv_encoder.alpha.grad``.  Only ``icv`` (already alpha-scaled, ref :89-92) carries
gradient through the LMM: the Functions below keep the explicit HIP forward/backward of ``licv.train_engine`` behind
autograd nodes — no torch ops run on the activations, and an incoming upstream gradient is read by the kernels from
device memory (no host sync).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops


def _rows2d(logits: torch.Tensor) -> torch.Tensor:
    """(B, S, V) logits with a (possibly padded) row stride -> the (B*S, V) strided view the row kernels index."""
    if logits.dim() == 2:
        return logits if logits.stride(1) == 1 else logits.contiguous()
    B, S, V = logits.shape
    if logits.stride(2) != 1 or logits.stride(0) != S * logits.stride(1):
        logits = logits.contiguous()
    return logits.as_strided((B * S, V), (logits.stride(1), 1))


def _padded_grad(shape, device):
    """Zero bf16 gradient buffer for logits of `shape`, rows padded to 8 columns; returns (2-D buffer, view of `shape`)."""
    V = shape[-1]
    rows = 1
    for d in shape[:-1]:
        rows *= int(d)
    ld = (V + 7) // 8 * 8
    buf = torch.zeros((rows, ld), dtype=torch.bfloat16, device=device)
    view = buf.as_strided(tuple(shape), (shape[1] * ld, ld, 1)) if len(shape) == 3 else buf[:, :V]
    return buf, view


class HookedStudentFn(torch.autograd.Function):
    """logits = LMM(inputs) with the ICV hook on `hook_layers`; backward = explicit HIP backward through the frozen LMM."""

    @staticmethod
    def forward(ctx, icv: torch.Tensor, student, inputs: dict, hook_layers: Sequence[int], logits_rows: Optional[torch.Tensor]):
        B, S = inputs["input_ids"].shape
        dev = student.e.w.device
        rows = logits_rows if logits_rows is not None else torch.arange(B * S, device=dev)
        logits, st = student.forward(**inputs, icv=icv.detach(), hook_layers=list(hook_layers), alpha=None, logits_rows=rows)
        ctx.student, ctx.st = student, st
        if logits_rows is None:
            V = logits.shape[-1]
            logits = logits.as_strided((B, S, V), (S * logits.stride(0), logits.stride(0), 1))
        return logits

    @staticmethod
    def backward(ctx, dlogits: torch.Tensor):
        student, st = ctx.student, ctx.st
        ctx.st = None                                               # the saved activations are consumed once
        V = dlogits.shape[-1]
        R = dlogits.numel() // V
        ld = (V + 7) // 8 * 8
        full = torch.zeros((R, ld), dtype=torch.bfloat16, device=dlogits.device)
        full[:, :V] = dlogits.reshape(R, V)
        grad_v = student.backward(st, full)                          # (1, n_hooked, H) fp32 = d loss / d (alpha*icv)
        return grad_v, None, None, None, None


class MaskedKLFn(torch.autograd.Function):
    """T^2 * mean_rows sum_v p (log(p+eps) - log(q+eps)) over the answer rows (ref:icv_src/icv_module.py:121-134), rows
    gathered by index from the student / teacher logits; gradient to the student logits only (the teacher runs under
    no_grad in the reference, :103-105)."""

    @staticmethod
    def forward(ctx, stu_logits, tea_logits, s_rows, t_rows, temperature: float, eps: float, t_param: Optional[torch.Tensor] = None):
        """`temperature` is the host value the kernels take; `t_param` is the module's temperature Parameter when it is
        trainable (`learnable_t`, ref:icv_src/icv_module.py:49-52) and only marks the node as differentiable in T."""
        s2, t2 = _rows2d(stu_logits), _rows2d(tea_logits)
        V = s2.shape[1]
        rows = ops.kl_rows(s2, t2, s_rows, t_rows, V, temperature, eps)
        ctx.save_for_backward(stu_logits, tea_logits, s_rows, t_rows)
        ctx.T, ctx.eps = temperature, eps
        ctx.t_param = t_param is not None and t_param.requires_grad
        ctx.t_like = t_param
        mean = rows.to(stu_logits.dtype).mean().float()
        ctx.kl_mean = mean if ctx.t_param else None
        return mean * (temperature * temperature)

    @staticmethod
    def backward(ctx, g):
        stu_logits, tea_logits, s_rows, t_rows = ctx.saved_tensors
        s2, t2 = _rows2d(stu_logits), _rows2d(tea_logits)
        V = s2.shape[1]
        g32 = g.detach().to(torch.float32).reshape(1).contiguous()
        d = ops.kl_rows_bwd(s2, t2, s_rows, t_rows, V, ctx.T, ctx.eps, upstream=1.0, upstream_dev=g32)
        buf, view = _padded_grad(stu_logits.shape, stu_logits.device)
        buf.index_copy_(0, s_rows, d)
        g_t = None
        if ctx.t_param:                                              # L = T^2 mean_rows f(T):  dL/dT = T^2 mean f' + 2 T mean f
            df = ops.kl_rows_dtemp(s2, t2, s_rows, t_rows, V, ctx.T, ctx.eps)
            g_t = (g32 * (ctx.T * ctx.T * df.mean() + 2.0 * ctx.T * ctx.kl_mean)).reshape(ctx.t_like.shape).to(ctx.t_like.device, ctx.t_like.dtype)
        return view, None, None, None, None, None, g_t


class ShiftedCEFn(torch.autograd.Function):
    """mean over the kept positions of CE(logits[:, t], labels[:, t+1]) (HF ForCausalLMLoss with pads masked)."""

    @staticmethod
    def forward(ctx, logits, rows, tokens):
        flat = _rows2d(logits)
        V = flat.shape[1]
        ctx.save_for_backward(logits, rows, tokens)
        return ops.ce_rows(flat, rows, tokens, V).mean()

    @staticmethod
    def backward(ctx, g):
        logits, rows, tokens = ctx.saved_tensors
        flat = _rows2d(logits)
        V = flat.shape[1]
        buf, view = _padded_grad(logits.shape, logits.device)
        ops.ce_rows(flat, rows, tokens, V, grad=buf, grad_coef=1.0 / max(int(rows.numel()), 1), grad_rows=rows, want_loss=False,
                    grad_coef_dev=g.detach().to(torch.float32).reshape(1).contiguous())
        return view, None, None
