"""Output record + base class of the ICV encoders (API of ref:icv_src/icv_encoder/base_icv_encoder.py:7-23)."""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class ICVEncoderOutput:
    """What an encoder hands to the intervention: the per-layer vectors and their scales."""
    in_context_feature: Optional[torch.Tensor]      # unused by the global encoder (always None)
    in_context_vector: Optional[torch.Tensor]       # (1, n_layers, hidden) fp32
    alpha: Optional[torch.Tensor]                   # (1, n_layers) fp32, post-sigmoid when enabled


class BaseICVEncoder(torch.nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.alpha = None
        self.icv_encoder = None

    def forward(self, *args, **kwargs) -> ICVEncoderOutput:
        raise NotImplementedError
