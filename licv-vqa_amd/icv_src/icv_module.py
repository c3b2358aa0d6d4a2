"""``VQAICVModule`` — the L-ICV distillation step behind the reference's surface, on the native engine.

Kept from ref:icv_src/icv_module.py:15-216: constructor ``(interface, module_cfg, lmm_cfg)``; sub-modules
``icv_model`` / ``icv_encoder`` (state-dict keys ``icv_encoder.alpha`` / ``icv_encoder.icv``); ``temperature``;
``forward(query_inputs, inputs, query_x_length, in_context_length) -> (loss_dict, ICVEncoderOutput)``;
``calculate_kl_divergence``; ``get_mask``; ``decay_temperature``; ``training_step``; ``configure_optimizers`` (two lr
groups, AdamW, cosine warm-up); ``on_save_checkpoint``.  PyTorch-Lightning / Hydra / DeepSpeed are not used: configs are
plain attribute namespaces (or dicts) with the reference's key names; ``licv.trainer.ICVTrainer`` is the data-parallel
loop, and a caller that drives the module the Lightning way (``loss = training_step(batch, i); loss.backward();
optimizer.step(); scheduler.step()``) gets the same numbers: the returned loss is differentiable w.r.t. ``icv`` / ``alpha``
through ``licv.autograd`` (explicit HIP backward behind autograd nodes).

Teacher and student forwards, the masked KL rows and AdamW run as HIP kernels.  CE is computed here with pads
masked by ``attention_mask`` (the pinned transformers 4.38.2 Idefics behaviour).
"""
from __future__ import annotations

import math
import types
from typing import Any

import torch

from licv import ops

from .icv_encoder.global_icv_encoder import GlobalICVEncoder
from .icv_model.icv_intervention import LearnableICVInterventionLMM


def _ns(cfg) -> Any:
    if isinstance(cfg, dict):
        return types.SimpleNamespace(**{k: _ns(v) if isinstance(v, dict) else v for k, v in cfg.items()})
    return cfg


def _get(cfg, key, default=None):
    return getattr(cfg, key, default) if not isinstance(cfg, dict) else cfg.get(key, default)


class VQAICVModule(torch.nn.Module):
    def __init__(self, interface, module_cfg, lmm_cfg) -> None:
        super().__init__()
        self.module_cfg, self.lmm_cfg = _ns(module_cfg), _ns(lmm_cfg)
        self.interface = interface
        self.interface.requires_grad_(False)
        if hasattr(self.interface.model, "gradient_checkpointing_enable"):
            self.interface.model.gradient_checkpointing_enable()
        self.icv_model = LearnableICVInterventionLMM(
            interface, enable_intervention=True, intervention_layer=self.lmm_cfg.intervention_layer,
            layer_format=self.lmm_cfg.layer_format, total_layers=self.lmm_cfg.total_layers)
        enc_cfg = _get(self.module_cfg, "icv_encoder", None) or {}
        enc_kw = {k: _get(enc_cfg, k) for k in ("alpha_learnable", "alpha_init_value", "use_sigmoid") if _get(enc_cfg, k) is not None}
        self.icv_encoder = GlobalICVEncoder(lmm_hidden_dim=self.lmm_cfg.hidden_size,
                                            lmm_layers=len(self.icv_model.intervention_layer_names), **enc_kw)
        self.temperature = torch.nn.Parameter(torch.tensor(float(self.module_cfg.init_temperature)),
                                              requires_grad=bool(_get(self.module_cfg, "learnable_t", False)))
        self.global_step = 0
        self.decay_per_step = None
        self._t_host = (None, -1)                   # (float value, version counter it was read at): see _temperature_value

    # ------------------------------------------------------------------ masks / losses
    def get_mask(self, inputs, mask_length):
        """mask[b,t] = (t >= length[b]) & (ids[b,t] != pad): assumes right padding (ref :136-148)."""
        ids = inputs[self.interface.input_ids_field_name]
        steps = torch.arange(ids.shape[1], device=ids.device).unsqueeze(0).expand(ids.shape[0], -1)
        return (steps >= mask_length.to(ids.device).unsqueeze(1)) & (ids != self.interface.tokenizer.pad_token_id)

    def _temperature_value(self) -> float:
        """The temperature as a host float for the kernels' scalar argument, read back only when the Parameter has changed
        (an optimiser step with `learnable_t`, or decay_temperature): the steady state makes no device round trip."""
        t = self.temperature
        ver = t._version
        if self._t_host[1] != ver or self._t_host[0] is None:
            self._t_host = (float(t.detach()), ver)
        return self._t_host[0]

    def calculate_kl_divergence(self, stu_logits, tea_logits):
        """mean over rows of sum_v p*(log(p+eps)-log(q+eps)), times T^2 (ref :121-134).  Rows are given as 2-D
        (rows, V) tensors; the per-row reduction over the vocabulary is one HIP kernel.  Differentiable like the reference's
        formula - in the student logits and, with ``learnable_t``, in the temperature - whenever autograd is recording
        (the same node as forward() uses, with identity row indices)."""
        n, V = stu_logits.shape
        idx = torch.arange(n, device=stu_logits.device)
        if torch.is_grad_enabled() and (stu_logits.requires_grad or self.temperature.requires_grad):
            return self._kl_from_rows(stu_logits, tea_logits.detach(), idx, idx)
        rows = ops.kl_rows(stu_logits if stu_logits.stride(1) == 1 else stu_logits.contiguous(),
                           tea_logits if tea_logits.stride(1) == 1 else tea_logits.contiguous(),
                           idx, idx, V, self._temperature_value(), float(self.module_cfg.kl_eps))
        return rows.to(stu_logits.dtype).mean() * self.temperature.detach().to(rows.device) ** 2

    def _kl_from_rows(self, stu_logits, tea_logits, s_rows, t_rows):
        """Same value as calculate_kl_divergence(stu[mask], tea[mask]) without materialising the gathered rows; differentiable
        w.r.t. the student logits and — with ``learnable_t`` (ref :49-52) — w.r.t. ``temperature`` (licv.autograd.MaskedKLFn:
        the kernels take the temperature as a host scalar, its gradient is one more per-row reduction, `licv_kl_rows_dtemp`)."""
        from licv.autograd import MaskedKLFn
        assert s_rows.numel() == t_rows.numel(), "student and teacher must mask the same number of answer tokens"
        t_param = self.temperature if self.temperature.requires_grad else None
        return MaskedKLFn.apply(stu_logits, tea_logits, s_rows, t_rows, self._temperature_value(), float(self.module_cfg.kl_eps), t_param)

    # ------------------------------------------------------------------ forward (ref :71-119)
    def forward(self, query_inputs, inputs, query_x_length, in_context_length):
        icl_context_mask = self.get_mask(inputs, in_context_length)
        zero_shot_mask = self.get_mask(query_inputs, query_x_length)
        enc = self.icv_encoder()
        icv = enc.alpha.unsqueeze(dim=-1) * enc.in_context_vector
        if self.module_cfg.hard_loss_weight:
            query_inputs["labels"] = query_inputs["input_ids"]
        self.icv_model.toggle_intervention(True)
        icv_outputs = self.icv_model(**query_inputs, icv=icv)
        if _get(self.module_cfg, "only_hard_loss", False):
            return {"loss": icv_outputs["loss"]}, enc
        s_rows = zero_shot_mask.reshape(-1).nonzero().squeeze(1)
        t_rows = icl_context_mask.reshape(-1).nonzero().squeeze(1)
        with torch.no_grad():
            self.icv_model.toggle_intervention(False)
            tea_kw = {k: v for k, v in inputs.items() if k != "labels"}
            if getattr(self.interface, "supports_logits_rows", False):
                # only the answer rows of the teacher enter the loss (ref :108-110): the LM head runs on those rows alone
                ice_logits = self.icv_model(**tea_kw, logits_rows=t_rows.to(self.interface.device))["logits"]
                t_rows = torch.arange(t_rows.numel(), device=ice_logits.device)
            else:
                ice_logits = self.icv_model(**tea_kw)["logits"]
        dev = icv_outputs["logits"].device
        kl_loss = self._kl_from_rows(icv_outputs["logits"], ice_logits, s_rows.to(dev), t_rows.to(dev))
        loss = 0.0 + kl_loss
        loss_dict = {"kl_loss": kl_loss}
        if self.module_cfg.hard_loss_weight:
            loss = loss + self.module_cfg.hard_loss_weight * icv_outputs["loss"]
            loss_dict["ce_loss"] = icv_outputs["loss"]
        loss_dict["loss"] = loss
        return loss_dict, enc

    # ------------------------------------------------------------------ Lightning-shaped hooks (ref :160-169, :171-216)
    def log_dict(self, values, **_):                 # Lightning's logger is replaced by a plain record of the last values
        self.logged = getattr(self, "logged", {})
        self.logged.update({k: (v.detach() if torch.is_tensor(v) else v) for k, v in values.items()})

    def log(self, name, value, **_):
        self.log_dict({name: value})

    def training_step(self, batch, batch_idx=0):
        self.decay_temperature()
        loss_dict, icv_encoder_output = self(**batch)
        self.log_dict(loss_dict, sync_dist=True, prog_bar=True)
        if _get(self.module_cfg, "log_alpha", False):
            alpha = icv_encoder_output.alpha.squeeze()
            for i in range(len(alpha)):
                self.log(f"alpha/alpha-{i}", alpha[i])
        self.log("temperature", self.temperature)
        return loss_dict["loss"]

    def configure_optimizers(self, estimated_stepping_batches: int = None):
        """Two lr groups (names containing "alpha" get alpha_lr), AdamW with weight decay, cosine schedule with warm-up stepped
        per optimiser step.  The optimiser is ``licv.optim.FusedAdamW`` (a ``torch.optim.Optimizer`` whose ``step`` is the HIP
        AdamW kernel; replaces torch AdamW / DeepSpeedCPUAdam, ref :181-192)."""
        from licv.optim import FusedAdamW
        params = []
        for name, param in self.icv_encoder.named_parameters():
            if not param.requires_grad:
                continue
            if "alpha" in name:
                params.append({"params": param, "lr": self.module_cfg.alpha_lr})
            else:
                params.append({"params": param})
        optimizer = FusedAdamW(params, lr=self.module_cfg.icv_lr, weight_decay=self.module_cfg.weight_decay)
        if estimated_stepping_batches is None:
            estimated_stepping_batches = self.trainer.estimated_stepping_batches
        spec = self.optimizer_spec(estimated_stepping_batches)
        warm, total = spec["warm_steps"], spec["total_steps"]
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda step: self.lr_lambda(step, warm, total))
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "interval": "step"}}

    def on_save_checkpoint(self, checkpoint):
        for name in list(checkpoint["state_dict"].keys()):       # only the icv weights are kept (ref :211-216)
            if name.startswith("model"):
                checkpoint["state_dict"].pop(name)

    # ------------------------------------------------------------------ schedule helpers (ref :54-69, :150-158, :171-209)
    def setup_temperature_decay(self, estimated_stepping_batches: int):
        d = self.module_cfg.decay_per_step
        if d < 0:
            return
        if isinstance(d, int):
            self.decay_per_step = d
        elif isinstance(d, float) and 0 < d < 1:
            self.decay_per_step = int(estimated_stepping_batches * d)
        else:
            raise ValueError("decay_ratio must be an int or a float between 0 and 1")

    def decay_temperature(self):
        if self.module_cfg.decay_ratio < 0:
            return
        if self.global_step % self.decay_per_step == 0 and self.global_step != 0:
            # (the reference re-binds the nn.Parameter attribute to a plain tensor here, which torch refuses; same value, in place)
            with torch.no_grad():
                self.temperature.copy_(torch.clip(self.temperature * self.module_cfg.decay_ratio, min=self.module_cfg.min_tmeprature))

    def optimizer_spec(self, estimated_stepping_batches: int):
        """The reference's recipe as plain numbers: alpha group lr, icv group lr, weight decay, warm-up steps."""
        w = self.module_cfg.warm_steps
        if isinstance(w, float):
            warm = w * estimated_stepping_batches
        elif isinstance(w, int):
            warm = w
        else:
            raise ValueError(f"the warm_steps should be int or float, but got {type(w)}")
        return dict(alpha_lr=float(self.module_cfg.alpha_lr), icv_lr=float(self.module_cfg.icv_lr),
                    weight_decay=float(self.module_cfg.weight_decay), warm_steps=warm, total_steps=estimated_stepping_batches)

    @staticmethod
    def lr_lambda(step: int, warm: float, total: float) -> float:
        """transformers.get_cosine_schedule_with_warmup (num_cycles 0.5)."""
        if step < warm:
            return float(step) / float(max(1, warm))
        progress = float(step - warm) / float(max(1, total - warm))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))
