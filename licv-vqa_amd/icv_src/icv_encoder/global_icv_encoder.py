"""The learnable global ICV: one vector and one scale per hooked layer.

API and initialisation order follow ref:icv_src/icv_encoder/global_icv_encoder.py:6-43 (alpha is created
before icv, icv ~ N(0, 0.01) from the global torch RNG), so a run seeded like the reference starts from
bit-identical parameters and its ``state_dict`` keys (``alpha``, ``icv``) load either way.
"""
import torch

from .base_icv_encoder import BaseICVEncoder, ICVEncoderOutput


class GlobalICVEncoder(BaseICVEncoder):
    def __init__(self, lmm_hidden_dim, lmm_layers, alpha_learnable=True, alpha_init_value=0.0, use_sigmoid=False) -> None:
        super().__init__()
        n_layers, hidden = int(lmm_layers), int(lmm_hidden_dim)
        scale = torch.full(size=(1, n_layers), fill_value=float(alpha_init_value))
        self.alpha = torch.nn.Parameter(scale, requires_grad=alpha_learnable)
        vectors = torch.empty(1, n_layers, hidden)
        torch.nn.init.normal_(vectors, mean=0.0, std=0.01)
        self.icv = torch.nn.Parameter(vectors)
        self.use_sigmoid = use_sigmoid

    def get_alpha(self):
        return self.alpha.sigmoid() if self.use_sigmoid else self.alpha

    def forward(self) -> ICVEncoderOutput:
        return ICVEncoderOutput(in_context_feature=None, in_context_vector=self.icv, alpha=self.get_alpha())

    def flat_parameters(self):
        """(alpha | icv) as the single fp32 buffer the fused AdamW / RCCL all-reduce operate on."""
        return torch.cat([self.alpha.detach().reshape(-1), self.icv.detach().reshape(-1)])
