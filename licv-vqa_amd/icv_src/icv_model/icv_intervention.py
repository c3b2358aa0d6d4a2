"""``LearnableICVInterventionLMM`` with the reference's surface, running on the native hook mechanism.

Surface kept from ref:icv_src/icv_model/icv_intervention.py:10-129: constructor arguments; attributes
``lmm, total_layers, intervention_layers, intervention_layer_names, layer_to_icv_index, intervention_enabled``
(absent when ``enable_intervention=False``, as in the reference); ``device``; ``intervention_status`` with its
``ValueError``; ``toggle_intervention``; ``forward(icv=None, *a, **kw)`` / ``generate(icv=None, *a, **kw)``.
What differs is underneath: no baukit, no retained or cloned activations (ref:README.md:14) — the edit is one
fused HIP kernel per hooked layer, inside the native engine's layer loop when ``lmm`` is a native interface.
One deliberate deviation: with ``enable_intervention=False`` the reference raises AttributeError on
forward/generate (``intervention_enabled`` is never set, ref :22 vs :89); here that case runs un-hooked,
which is what ref:inference.py:110 (the ICL baseline) needs.
"""
from contextlib import nullcontext
from typing import List, Union

import torch.nn as nn

from licv.intervention import NativeIntervention


class LearnableICVInterventionLMM(nn.Module):
    def __init__(self, lmm: nn.Module, enable_intervention=True, intervention_layer: Union[int, List[int]] = None,
                 layer_format: str = None, total_layers: int = None):
        super().__init__()
        self.lmm = lmm
        if not enable_intervention:
            return
        self.total_layers = total_layers
        self.intervention_layers = self._prepare_layers(intervention_layer)
        self.intervention_layer_names = [layer_format.replace("<LAYER_NUM>", str(i)) for i in self.intervention_layers]
        self.layer_to_icv_index = {int(layer): int(slot) for slot, layer in enumerate(self.intervention_layers)}
        self.intervention_enabled = True

    def _prepare_layers(self, layers):
        if layers == -1:
            return list(range(self.total_layers))
        if isinstance(layers, int):
            return [layers]
        return layers

    @property
    def device(self):
        return self.lmm.device

    @property
    def intervention_status(self) -> bool:
        return self.intervention_enabled

    @intervention_status.setter
    def intervention_status(self, value: bool):
        if not isinstance(value, bool):
            raise ValueError("Intervention status must be a boolean value.")
        self.intervention_enabled = value

    def toggle_intervention(self, enable: bool):
        self.intervention_status = enable

    def _get_context_manager(self, icv=None, retain_grad=False):
        if getattr(self, "intervention_enabled", False):
            return NativeIntervention(self.lmm, self.intervention_layer_names, self.layer_to_icv_index, icv,
                                      retain_grad=retain_grad)
        return nullcontext()

    def forward(self, icv=None, *args, **kwargs):
        """``icv``: (1, n_hooked, hidden), already alpha-scaled (ref:icv_src/icv_module.py:89-92); the rest goes to the LMM."""
        with self._get_context_manager(icv, retain_grad=True):
            return self.lmm(*args, **kwargs)

    def generate(self, icv=None, *args, **kwargs):
        with self._get_context_manager(icv, retain_grad=False):
            return self.lmm.generate(*args, **kwargs)
