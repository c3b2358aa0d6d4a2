"""Loading a real checkpoint directory (ref:utils.py:42,70: ``model_name_or_path = MODEL_CPK_DIR/<model_name>``; weights come from
``huggingface-cli download``, ref:README.md:33-40 — none exist offline).  A tiny ``save_pretrained`` directory written by the
installed transformers stands in: the loader must recover the architecture from config.json and a state dict whose names are the
ones the native engines index (the same names licv.synthetic generates), with identical tensors."""
import pytest
import torch

from licv.config import IDEFICS2_TINY, IDEFICS_TINY, Idefics2Arch, IdeficsArch
from licv.synthetic import synth_idefics2_weights, synth_idefics_weights

transformers = pytest.importorskip("transformers")


def _save_idefics(tmp_path, arch, sd):
    from transformers import IdeficsConfig, IdeficsForVisionText2Text
    cfg = IdeficsConfig(
        vocab_size=arch.vocab_size, additional_vocab_size=arch.additional_vocab_size, hidden_size=arch.hidden_size,
        intermediate_size=arch.intermediate_size, num_hidden_layers=arch.num_layers, num_attention_heads=arch.num_heads,
        rms_norm_eps=arch.rms_eps, cross_layer_interval=arch.cross_layer_interval, qk_layer_norms=arch.qk_layer_norms,
        use_resampler=arch.use_resampler, alpha_initializer="ones", alpha_type="float", pad_token_id=arch.pad_token_id,
        bos_token_id=arch.bos_token_id, eos_token_id=arch.eos_token_id,
        vision_config=dict(embed_dim=arch.v_embed, image_size=arch.v_image, patch_size=arch.v_patch, num_hidden_layers=arch.v_layers,
                           num_attention_heads=arch.v_heads, intermediate_size=arch.v_inter, layer_norm_eps=arch.v_ln_eps, hidden_act=arch.v_act),
        perceiver_config=dict(use_resampler=arch.use_resampler, resampler_n_latents=arch.r_latents, resampler_depth=arch.r_depth,
                              resampler_n_heads=arch.r_heads, resampler_head_dim=arch.r_head_dim, qk_layer_norms_perceiver=arch.r_qk_norm))
    m = IdeficsForVisionText2Text(cfg)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary_emb" in k or "position_ids" in k for k in missing)
    m.save_pretrained(tmp_path)


def test_idefics_checkpoint_directory_round_trip(tmp_path):
    from lmm_icl_interface.interface import IdeficsInterface
    arch = IDEFICS_TINY
    sd = synth_idefics_weights(arch, seed=9, dtype=torch.float32)
    _save_idefics(tmp_path, arch, sd)
    got_sd, got_arch, _, _ = IdeficsInterface._load_checkpoint(tmp_path, None, None)
    assert isinstance(got_arch, IdeficsArch) and got_arch == arch
    for k, v in sd.items():
        if k.startswith("model.vision_model.post_layernorm"):               # not on the path (pooled output only)
            continue
        assert k in got_sd, f"{k} missing from the loaded state dict"
        assert torch.equal(got_sd[k].float(), v), k
    with pytest.raises(FileNotFoundError):
        IdeficsInterface._load_checkpoint(tmp_path / "nope", None, None)


def test_idefics2_checkpoint_directory_round_trip(tmp_path):
    from transformers import Idefics2Config, Idefics2ForConditionalGeneration
    from lmm_icl_interface.interface import Idefics2Interface
    arch = IDEFICS2_TINY
    sd = synth_idefics2_weights(arch, seed=9, dtype=torch.float32)
    cfg = Idefics2Config(
        vision_config=dict(hidden_size=arch.v_hidden, intermediate_size=arch.v_inter, num_hidden_layers=arch.v_layers,
                           num_attention_heads=arch.v_heads, image_size=arch.v_image, patch_size=arch.v_patch, hidden_act=arch.v_act,
                           layer_norm_eps=arch.v_ln_eps),
        perceiver_config=dict(hidden_size=arch.hidden_size, resampler_n_latents=arch.r_latents, resampler_depth=arch.r_depth,
                              resampler_n_heads=arch.r_heads, resampler_head_dim=arch.r_head_dim, num_key_value_heads=arch.r_kv_heads,
                              hidden_act="silu", rms_norm_eps=arch.rms_eps),
        text_config=dict(model_type="mistral", vocab_size=arch.vocab_size, hidden_size=arch.hidden_size, intermediate_size=arch.intermediate_size,
                         num_hidden_layers=arch.num_layers, num_attention_heads=arch.num_heads, num_key_value_heads=arch.num_kv_heads,
                         rms_norm_eps=arch.rms_eps, max_position_embeddings=4096, sliding_window=4096, pad_token_id=arch.pad_token_id,
                         rope_parameters=dict(rope_type="default", rope_theta=arch.rope_base)),
        image_token_id=arch.image_token_id)
    m = Idefics2ForConditionalGeneration(cfg)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not missing
    m.save_pretrained(tmp_path)
    got_sd, got_arch, _, _ = Idefics2Interface._load_checkpoint(tmp_path, None, None, Idefics2Arch)
    assert isinstance(got_arch, Idefics2Arch)
    for f in ("vocab_size", "hidden_size", "intermediate_size", "num_layers", "num_heads", "num_kv_heads", "image_token_id", "v_hidden",
              "v_inter", "v_layers", "v_heads", "v_image", "v_patch", "r_latents", "r_depth", "r_heads", "r_head_dim", "r_kv_heads"):
        assert getattr(got_arch, f) == getattr(arch, f), f
    for k, v in sd.items():
        assert k in got_sd and torch.equal(got_sd[k].float(), v), k
