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
import re
from contextlib import nullcontext
from typing import List, Union

import torch.nn as nn

from licv.intervention import ICVHookFn, NativeIntervention


def _layer_list(spec: Union[int, List[int]], total: int) -> List[int]:
    """-1 -> every layer; an int -> that layer; a list -> itself (ref:icv_src/icv_model/icv_intervention.py:36-43)."""
    if isinstance(spec, int):
        return list(range(total)) if spec == -1 else [spec]
    return spec


class LearnableICVInterventionLMM(nn.Module):
    def __init__(self, lmm: nn.Module, enable_intervention=True, intervention_layer: Union[int, List[int]] = None,
                 layer_format: str = None, total_layers: int = None):
        super().__init__()
        self.lmm = lmm
        if enable_intervention:
            self.total_layers = total_layers
            self.intervention_layers = _layer_list(intervention_layer, total_layers)
            self.intervention_layer_names = [layer_format.replace("<LAYER_NUM>", str(i)) for i in self.intervention_layers]
            self.layer_to_icv_index = {int(layer): int(slot) for slot, layer in enumerate(self.intervention_layers)}
            self.intervention_enabled = True

    def _prepare_layers(self, layers):                      # kept: callers of the reference's private helper
        return _layer_list(layers, self.total_layers)

    device = property(lambda self: self.lmm.device)

    def _get_status(self) -> bool:
        return self.intervention_enabled

    def _set_status(self, value: bool) -> None:
        if type(value) is not bool:
            raise ValueError("Intervention status must be a boolean value.")
        self.intervention_enabled = value

    intervention_status = property(_get_status, _set_status)

    def toggle_intervention(self, enable: bool):
        self._set_status(enable)

    def apply_icv_intervention(self, edit_layers, icv):
        """The reference's edit-function factory (ref :61-86): returns ``fn(output, layer_name)`` that applies
        h' = (h+v)/||h+v||*||h|| with v = icv[:, layer_to_icv_index[layer]] to a tensor output, or to element 0 of a tuple
        output, for layer names in ``edit_layers``; other outputs pass through.  The arithmetic is the fused HIP kernel
        (differentiable w.r.t. h and icv); outputs must live on the GPU."""
        names = set(edit_layers)

        def intervention_function(output, layer_name):
            if layer_name not in names:
                return output
            slot = self.layer_to_icv_index[int(re.findall(r"\d+", layer_name)[0])]
            if isinstance(output, tuple):
                hidden_states, *rest = output
                return (ICVHookFn.apply(hidden_states, icv, slot),) + tuple(rest)
            return ICVHookFn.apply(output, icv, slot)

        return intervention_function

    def _get_context_manager(self, icv=None, retain_grad=False):
        if not getattr(self, "intervention_enabled", False):
            return nullcontext()
        return NativeIntervention(self.lmm, self.intervention_layer_names, self.layer_to_icv_index, icv, retain_grad=retain_grad)

    def _run(self, fn, icv, retain_grad, args, kwargs):
        with self._get_context_manager(icv, retain_grad=retain_grad):
            return fn(*args, **kwargs)

    def forward(self, icv=None, *args, **kwargs):
        """``icv``: (1, n_hooked, hidden), already alpha-scaled (ref:icv_src/icv_module.py:89-92); the rest goes to the LMM."""
        return self._run(self.lmm, icv, True, args, kwargs)

    def generate(self, icv=None, *args, **kwargs):
        return self._run(self.lmm.generate, icv, False, args, kwargs)
