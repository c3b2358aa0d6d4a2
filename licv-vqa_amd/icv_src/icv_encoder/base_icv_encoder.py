"""Output record + base class of the ICV encoders (API of ref:icv_src/icv_encoder/base_icv_encoder.py:7-23)."""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class ICVEncoderOutput:
    """What an encoder hands to the intervention: a mutable dataclass with the reference's three fields, in its order
    (positional construction and attribute assignment both behave as there)."""
    in_context_feature: Optional[torch.Tensor]      # unused by the global encoder
    in_context_vector: Optional[torch.Tensor]       # (1, n_layers, hidden) fp32
    alpha: Optional[torch.Tensor]                   # (1, n_layers) fp32, post-sigmoid when enabled


class BaseICVEncoder(torch.nn.Module):
    """Encoders own ``alpha`` (per-layer scale) and whatever produces the vectors; subclasses implement ``forward``."""

    def __init__(self) -> None:
        super().__init__()
        for slot in ("alpha", "icv_encoder"):
            setattr(self, slot, None)

    def forward(self, *args, **kwargs) -> ICVEncoderOutput:
        raise NotImplementedError(f"{type(self).__name__} does not implement forward()")
