"""``FusedAdamW``: a ``torch.optim.Optimizer`` whose update is the HIP AdamW kernel (``licv_adamw_step``).

Stands in for ``torch.optim.AdamW`` / ``DeepSpeedCPUAdam`` in ``VQAICVModule.configure_optimizers``
(ref:icv_src/icv_module.py:171-192): same hyper-parameter names, decoupled weight decay, bias correction by step count;
torch's LR schedulers drive it through ``param_groups[i]["lr"]`` as usual.  Each parameter keeps fp32 ``exp_avg`` /
``exp_avg_sq`` state on its own device; the whole update of a parameter is one kernel launch (no host round trip — the
reference's ZeRO-2 offload moves these 131 k floats to the host and back every step).
"""
from __future__ import annotations

import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                assert p.dtype == torch.float32 and p.is_contiguous(), "FusedAdamW holds fp32 contiguous parameters"
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                lr = float(group["lr"])
                ops.adamw_step_(p.view(-1), p.grad.to(torch.float32).contiguous().view(-1), st["exp_avg"].view(-1),
                                st["exp_avg_sq"].view(-1), 0, lr, lr, st["step"], beta1=b1, beta2=b2, eps=group["eps"],
                                weight_decay=group["weight_decay"], grad_scale=1.0)
        return loss
