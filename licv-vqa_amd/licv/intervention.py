"""The hook mechanism, natively: replaces baukit.TraceDict for the L-ICV path
(ref:icv_src/icv_model/icv_intervention.py:88-98; baukit semantics in SURVEY.md §8 a5).

Two back ends behind one context manager:
  * a native LMM interface (``licv`` engines) receives the (layer -> icv row) plan and fuses the hook into its
    own layer loop — no module hooks, no retained/cloned activations;
  * any other ``torch.nn.Module`` on the GPU (e.g. an HF model) gets ``register_forward_hook`` handles whose
    edit is the fused HIP kernel, wrapped in an autograd Function so gradients reach ``icv``.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional

import torch

from . import ops


class ICVHookFn(torch.autograd.Function):
    """h' = (h+v)/||h+v||*||h|| with v = icv[0, idx]; backward through the HIP kernel."""

    @staticmethod
    def forward(ctx, h: torch.Tensor, icv: torch.Tensor, idx: int):
        hc = h.contiguous()
        row = icv[0, idx].contiguous().float()
        ctx.save_for_backward(hc, row)
        ctx.idx, ctx.icv_shape, ctx.h_dtype = idx, icv.shape, h.dtype
        return ops.inject_renorm(hc, row)

    @staticmethod
    def backward(ctx, grad_out):
        hc, row = ctx.saved_tensors
        gh, gv = ops.inject_renorm_bwd(hc, row, None, grad_out.contiguous().float(), need_grad_h=ctx.needs_input_grad[0])
        gi = None
        if ctx.needs_input_grad[1]:
            gi = torch.zeros(ctx.icv_shape, dtype=torch.float32, device=hc.device)
            gi[0, ctx.idx] = gv
        return (gh.to(ctx.h_dtype) if gh is not None else None), gi, None


class NativeIntervention:
    """Context manager with TraceDict's call shape: ``with NativeIntervention(lmm, names, index, icv): lmm(...)``."""

    def __init__(self, lmm, layer_names: List[str], layer_to_icv_index: Dict[int, int], icv: Optional[torch.Tensor],
                 retain_grad: bool = False):
        self.lmm, self.names, self.index, self.icv = lmm, list(layer_names), dict(layer_to_icv_index), icv
        self._handles = []

    def __enter__(self):
        if self.icv is None:
            raise ValueError("intervention is enabled but no `icv` was passed")
        if hasattr(self.lmm, "install_intervention"):
            self.lmm.install_intervention(self.names, self.index, self.icv)
            return self
        modules = dict(self.lmm.named_modules())
        for name in self.names:
            if name not in modules:
                raise LookupError(f"no submodule named {name!r} to intervene on")
            layer_idx = int(re.findall(r"\d+", name)[0])                   # ref :63
            slot = self.index[layer_idx]

            def edit(mod, inputs, output, slot=slot):
                if isinstance(output, tuple):                              # ref :64-73 (transformers 4.38 layers)
                    return (ICVHookFn.apply(output[0], self.icv, slot),) + tuple(output[1:])
                return ICVHookFn.apply(output, self.icv, slot)             # ref :74-83
            self._handles.append(modules[name].register_forward_hook(edit))
        return self

    def __exit__(self, *exc):
        if hasattr(self.lmm, "remove_intervention"):
            self.lmm.remove_intervention()
        for h in self._handles:
            h.remove()
        self._handles = []
        return False
