"""Architecture descriptions for the LMMs the L-ICV hot path runs on.

Values are *read from* the HF config object of a checkpoint when one exists
(`IdeficsArch.from_hf`); the presets below restate the public Idefics-9B
numbers listed in SURVEY.md §8 so that the bench can build a random-init model of
the right shape without any checkpoint or network access.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict, replace


@dataclass(frozen=True)
class IdeficsArch:
    # language model (hf:idefics/configuration_idefics.py:126-151)
    vocab_size: int = 32000
    additional_vocab_size: int = 2
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_layers: int = 32
    num_heads: int = 32
    rms_eps: float = 1e-6
    cross_layer_interval: int = 4
    qk_layer_norms: bool = True
    max_positions: int = 2048
    rope_base: float = 10000.0
    pad_token_id: int = 0
    bos_token_id: int = 1
    eos_token_id: int = 2
    # vision tower (hf:idefics/configuration_idefics.py:33-44)
    v_embed: int = 1280
    v_image: int = 224
    v_patch: int = 14
    v_layers: int = 32
    v_heads: int = 16
    v_inter: int = 5120
    v_ln_eps: float = 1e-5
    v_act: str = "gelu"
    # perceiver resampler (hf:idefics/configuration_idefics.py:67-72)
    use_resampler: bool = True
    r_latents: int = 64
    r_depth: int = 6
    r_heads: int = 16
    r_head_dim: int = 96
    r_qk_norm: bool = True

    # ---- derived ----
    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def v_head_dim(self) -> int:
        return self.v_embed // self.v_heads

    @property
    def v_tokens(self) -> int:
        return (self.v_image // self.v_patch) ** 2 + 1

    @property
    def image_seq_len(self) -> int:
        return self.r_latents if self.use_resampler else self.v_tokens

    @property
    def num_cross_layers(self) -> int:
        return self.num_layers // self.cross_layer_interval

    @property
    def total_vocab(self) -> int:
        return self.vocab_size + self.additional_vocab_size

    def to_dict(self):
        return asdict(self)

    def with_(self, **kw) -> "IdeficsArch":
        return replace(self, **kw)

    @staticmethod
    def from_hf(cfg) -> "IdeficsArch":
        """Build from a transformers ``IdeficsConfig`` (never hard-code checkpoint values)."""
        v, p = cfg.vision_config, cfg.perceiver_config
        return IdeficsArch(
            vocab_size=cfg.vocab_size, additional_vocab_size=cfg.additional_vocab_size,
            hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
            num_layers=cfg.num_hidden_layers, num_heads=cfg.num_attention_heads,
            rms_eps=cfg.rms_norm_eps, cross_layer_interval=cfg.cross_layer_interval,
            qk_layer_norms=cfg.qk_layer_norms,
            pad_token_id=cfg.pad_token_id if cfg.pad_token_id is not None else 0,
            bos_token_id=cfg.bos_token_id, eos_token_id=cfg.eos_token_id,
            v_embed=v.embed_dim, v_image=v.image_size, v_patch=v.patch_size,
            v_layers=v.num_hidden_layers, v_heads=v.num_attention_heads,
            v_inter=v.intermediate_size, v_ln_eps=v.layer_norm_eps, v_act=v.hidden_act,
            use_resampler=cfg.use_resampler, r_latents=p.resampler_n_latents,
            r_depth=p.resampler_depth, r_heads=p.resampler_n_heads,
            r_head_dim=p.resampler_head_dim, r_qk_norm=p.qk_layer_norms_perceiver,
        )


IDEFICS_9B = IdeficsArch()

# Small shapes for parity tests.  "tiny" has toy head dims; "mid" keeps the real
# head dims of Idefics-9B (128 / 80 / 96) so every kernel specialisation is hit.
IDEFICS_TINY = IdeficsArch(
    vocab_size=96, additional_vocab_size=2, hidden_size=64, intermediate_size=128,
    num_layers=4, num_heads=4, cross_layer_interval=2,
    v_embed=32, v_image=28, v_patch=14, v_layers=2, v_heads=4, v_inter=64,
    r_latents=4, r_depth=2, r_heads=2, r_head_dim=8,
)
IDEFICS_MID = IdeficsArch(
    vocab_size=160, additional_vocab_size=2, hidden_size=256, intermediate_size=352,
    num_layers=4, num_heads=2, cross_layer_interval=2,
    v_embed=160, v_image=42, v_patch=14, v_layers=2, v_heads=2, v_inter=224,
    r_latents=8, r_depth=2, r_heads=2, r_head_dim=96,
)

@dataclass(frozen=True)
class Idefics2Arch:
    """Idefics2 (SigLIP NaViT tower + modality projection + GQA perceiver + Mistral); values from
    hf:idefics2/configuration_idefics2.py and the public idefics2-8b-base config."""
    # Mistral text model
    vocab_size: int = 32003
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_layers: int = 32
    num_heads: int = 32
    num_kv_heads: int = 8
    rms_eps: float = 1e-5
    rope_base: float = 10000.0
    pad_token_id: int = 0
    bos_token_id: int = 1
    eos_token_id: int = 2
    image_token_id: int = 32001
    # SigLIP vision tower
    v_hidden: int = 1152
    v_inter: int = 4304
    v_layers: int = 27
    v_heads: int = 16
    v_image: int = 980
    v_patch: int = 14
    v_ln_eps: float = 1e-6
    v_act: str = "gelu_pytorch_tanh"
    # perceiver resampler
    r_latents: int = 64
    r_depth: int = 3
    r_heads: int = 16
    r_head_dim: int = 96
    r_kv_heads: int = 4

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def v_head_dim(self) -> int:
        return self.v_hidden // self.v_heads

    @property
    def image_seq_len(self) -> int:
        return self.r_latents

    def with_(self, **kw) -> "Idefics2Arch":
        return replace(self, **kw)

    def to_dict(self):
        return asdict(self)

    @staticmethod
    def from_hf(cfg) -> "Idefics2Arch":
        t, v, p = cfg.text_config, cfg.vision_config, cfg.perceiver_config
        rope = getattr(t, "rope_parameters", None) or {}
        return Idefics2Arch(
            vocab_size=t.vocab_size, hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
            num_layers=t.num_hidden_layers, num_heads=t.num_attention_heads, num_kv_heads=t.num_key_value_heads,
            rms_eps=t.rms_norm_eps, rope_base=float(rope.get("rope_theta", getattr(t, "rope_theta", 10000.0))),
            pad_token_id=t.pad_token_id if t.pad_token_id is not None else 0, image_token_id=cfg.image_token_id,
            v_hidden=v.hidden_size, v_inter=v.intermediate_size, v_layers=v.num_hidden_layers, v_heads=v.num_attention_heads,
            v_image=v.image_size, v_patch=v.patch_size, v_ln_eps=v.layer_norm_eps, v_act=v.hidden_act,
            r_latents=p.resampler_n_latents, r_depth=p.resampler_depth, r_heads=p.resampler_n_heads,
            r_head_dim=p.resampler_head_dim, r_kv_heads=p.num_key_value_heads)


IDEFICS2_8B = Idefics2Arch()
IDEFICS2_TINY = Idefics2Arch(vocab_size=100, hidden_size=128, intermediate_size=192, num_layers=3, num_heads=4, num_kv_heads=2,
                             image_token_id=99, v_hidden=64, v_inter=96, v_layers=2, v_heads=4, v_image=56, v_patch=14,
                             r_latents=4, r_depth=2, r_heads=4, r_head_dim=16, r_kv_heads=2)
# real head dims (128 text / 72 vision / 96 perceiver), ragged images, GQA 4:1
IDEFICS2_MID = Idefics2Arch(vocab_size=160, hidden_size=256, intermediate_size=352, num_layers=3, num_heads=2, num_kv_heads=1,
                            image_token_id=159, v_hidden=144, v_inter=224, v_layers=2, v_heads=2, v_image=84, v_patch=14,
                            r_latents=8, r_depth=2, r_heads=4, r_head_dim=96, r_kv_heads=1)

PRESETS = {"idefics2-8b": IDEFICS2_8B, "idefics2-tiny": IDEFICS2_TINY, "idefics2-mid": IDEFICS2_MID, "idefics-9b": IDEFICS_9B, "idefics-tiny": IDEFICS_TINY, "idefics-mid": IDEFICS_MID}


def idefics_arch(name: str) -> IdeficsArch:
    return PRESETS[name]
