#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

What runs here is the reference itself, not a restatement:
  * ``icv_src.icv_encoder.global_icv_encoder.GlobalICVEncoder`` imported as-is from
    /root/reference;
  * ``icv_src.icv_model.icv_intervention.LearnableICVInterventionLMM`` imported as-is, with
    two absent third-party modules satisfied in ``sys.modules``: ``loguru`` (a logger that
    discards) and ``baukit`` (a ``TraceDict`` written here on ``register_forward_hook`` with
    baukit's published ``edit_output`` / ``retain_grad`` behaviour);
  * ``VQAICVModule.forward / calculate_kl_divergence / get_mask`` compiled from the method
    bodies found (via ``ast``) in /root/reference/icv_src/icv_module.py and bound to a plain
    ``nn.Module`` — the file itself cannot be imported (lightning/hydra/deepspeed absent);
  * the LMM arithmetic is the installed ``transformers`` Idefics model on tiny random-init
    configs, ``attn_implementation="eager"`` (weights from ``licv.synthetic``, seeded).

Only inputs and expected outputs are written (npz).  Nothing from /root/reference is copied.
"""
from __future__ import annotations

import ast
import contextlib
import os
import sys
import types
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "licv-vqa_amd"))
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"

from licv.config import IDEFICS_TINY, IDEFICS_MID, IDEFICS2_TINY, IDEFICS2_MID, IdeficsArch  # noqa: E402
from licv.synthetic import (synth_idefics_weights, synth_vqa_batch, weights_checksum, synth_idefics2_weights,  # noqa: E402
                            synth_vqa_batch_idefics2)

# The build's own drop-in package is also called ``icv_src``; make sure the name resolves to the
# REFERENCE here: drop the build's package dir from the path now that ``licv`` is imported.
sys.path.remove(str(ROOT / "licv-vqa_amd"))


# ----------------------------------------------------------------------------- shims
def _install_shims():
    lg = types.ModuleType("loguru")

    class _Quiet:
        def __getattr__(self, _):
            return lambda *a, **k: None
    lg.logger = _Quiet()
    sys.modules["loguru"] = lg

    bk = types.ModuleType("baukit")

    def _resolve(model, name):
        for n, m in model.named_modules():
            if n == name:
                return m
        raise LookupError(name)

    class _Slot:
        output = None

    class TraceDict(OrderedDict, contextlib.AbstractContextManager):
        """register_forward_hook per named layer; hook(output) -> edit_output(output, layer_name);
        retain_output keeps the edited output; retain_grad => .retain_grad() + return a clone."""

        def __init__(self, module, layers=None, retain_output=True, retain_grad=False, edit_output=None, **_):
            super().__init__()
            self._handles = []
            for name in layers:
                slot = _Slot()
                self[name] = slot

                def hook(mod, inp, out, name=name, slot=slot):
                    if edit_output is not None:
                        out = edit_output(out, name)
                    if retain_output:
                        slot.output = out
                        if retain_grad:
                            first = out[0] if isinstance(out, tuple) else out
                            if first.requires_grad:
                                first.retain_grad()
                            out = tuple(o.clone() if torch.is_tensor(o) else o for o in out) if isinstance(out, tuple) else out.clone()
                    return out
                self._handles.append(_resolve(module, name).register_forward_hook(hook))

        def __exit__(self, *exc):
            for h in self._handles:
                h.remove()
    bk.TraceDict = TraceDict
    sys.modules["baukit"] = bk


def _reference_module_methods():
    """Compile forward/calculate_kl_divergence/get_mask from the reference file's AST."""
    src = (REF / "icv_src" / "icv_module.py").read_text()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "VQAICVModule")
    wanted = {"forward", "calculate_kl_divergence", "get_mask"}
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    mod = ast.Module(body=fns, type_ignores=[])
    ns = {"torch": torch}
    exec(compile(mod, str(REF / "icv_src" / "icv_module.py"), "exec"), ns)
    return {k: ns[k] for k in wanted}


# ----------------------------------------------------------------------------- HF model
def hf_model(arch: IdeficsArch, sd, dtype):
    from transformers import IdeficsConfig, IdeficsForVisionText2Text
    cfg = IdeficsConfig(
        vocab_size=arch.vocab_size, additional_vocab_size=arch.additional_vocab_size,
        hidden_size=arch.hidden_size, intermediate_size=arch.intermediate_size,
        num_hidden_layers=arch.num_layers, num_attention_heads=arch.num_heads,
        rms_norm_eps=arch.rms_eps, cross_layer_interval=arch.cross_layer_interval,
        qk_layer_norms=arch.qk_layer_norms, use_resampler=arch.use_resampler,
        alpha_initializer="ones", alpha_type="float", pad_token_id=arch.pad_token_id,
        bos_token_id=arch.bos_token_id, eos_token_id=arch.eos_token_id,
        vision_config=dict(embed_dim=arch.v_embed, image_size=arch.v_image, patch_size=arch.v_patch,
                           num_hidden_layers=arch.v_layers, num_attention_heads=arch.v_heads,
                           intermediate_size=arch.v_inter, layer_norm_eps=arch.v_ln_eps, hidden_act=arch.v_act),
        perceiver_config=dict(use_resampler=arch.use_resampler, resampler_n_latents=arch.r_latents,
                              resampler_depth=arch.r_depth, resampler_n_heads=arch.r_heads,
                              resampler_head_dim=arch.r_head_dim, qk_layer_norms_perceiver=arch.r_qk_norm),
        attn_implementation="eager",
    )
    m = IdeficsForVisionText2Text(cfg)
    missing, unexpected = m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(("rotary_emb" in k or "position_ids" in k) for k in missing), missing
    return m.to(dtype).eval()


class Interface(torch.nn.Module):
    """Minimal stand-in for lmm_icl_interface.LMMInterface: .model + passthrough call/generate."""
    input_ids_field_name = "input_ids"

    def __init__(self, model, pad_token_id):
        super().__init__()
        self.model = model
        self.tokenizer = types.SimpleNamespace(pad_token_id=pad_token_id)

    @property
    def device(self):
        return next(self.model.parameters()).device

    def forward(self, **kw):
        return self.model(**kw)

    def generate(self, **kw):
        return self.model.generate(**kw)


def np_(t):
    t = t.detach()
    return t.float().numpy().copy() if t.dtype == torch.bfloat16 else t.numpy().copy()


# ----------------------------------------------------------------------------- fixtures
def g1_encoder():
    from icv_src.icv_encoder.global_icv_encoder import GlobalICVEncoder
    out = {}
    for tag, (H, L, a0, sig) in {"a": (64, 4, 0.0, False), "b": (48, 3, 0.3, True)}.items():
        torch.manual_seed(426)
        enc = GlobalICVEncoder(lmm_hidden_dim=H, lmm_layers=L, alpha_init_value=a0, use_sigmoid=sig)
        o = enc()
        out[f"{tag}_cfg"] = np.array([H, L, a0, float(sig)])
        out[f"{tag}_icv"] = np_(o.in_context_vector)
        out[f"{tag}_alpha_param"] = np_(enc.alpha)
        out[f"{tag}_alpha_out"] = np_(o.alpha)
        assert o.in_context_feature is None
        out[f"{tag}_state_keys"] = np.array(sorted(enc.state_dict().keys()))
    np.savez_compressed(OUT / "g1_encoder.npz", **out)


def g2_intervention():
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    g = torch.Generator().manual_seed(1)
    out = {}
    lmm = torch.nn.Linear(2, 2)
    w = LearnableICVInterventionLMM(lmm, enable_intervention=True, intervention_layer=[3, 7, 1],
                                    layer_format="model.model.layers.<LAYER_NUM>", total_layers=8)
    out["names"] = np.array(w.intervention_layer_names)
    out["map_keys"] = np.array(list(w.layer_to_icv_index.keys()))
    out["map_vals"] = np.array(list(w.layer_to_icv_index.values()))
    w_all = LearnableICVInterventionLMM(lmm, True, -1, "blk.<LAYER_NUM>.mlp", 5)
    out["names_all"] = np.array(w_all.intervention_layer_names)
    w_int = LearnableICVInterventionLMM(lmm, True, 2, "blk.<LAYER_NUM>", 5)
    out["names_int"] = np.array(w_int.intervention_layer_names)
    icv = torch.randn(1, 3, 96, generator=g) * 0.3
    fn = w.apply_icv_intervention(w.intervention_layer_names, icv)
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        h = (torch.randn(2, 5, 96, generator=g) * 2).to(dt)
        out[f"h_{dt_name}"] = np_(h)
        r = fn(h, "model.model.layers.7")
        out[f"tensor_{dt_name}"] = np_(r)
        out[f"tensor_{dt_name}_is_f32"] = np.array(r.dtype == torch.float32)
        rt = fn((h, "aux", 3), "model.model.layers.1")
        assert isinstance(rt, tuple) and rt[1:] == ("aux", 3)
        out[f"tuple_{dt_name}"] = np_(rt[0])
        same = fn(h, "model.model.layers.5")           # not an edited layer -> untouched
        assert same is h
    out["icv"] = np_(icv)
    # toggle semantics
    try:
        w.toggle_intervention(1)
        raise AssertionError("expected ValueError")
    except ValueError as e:
        out["toggle_error"] = np.array(str(e))
    np.savez_compressed(OUT / "g2_intervention.npz", **out)


def _run_idefics(arch, tag, seed, B, S, N, min_len, hook_sets):
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    out = {}
    sd32 = synth_idefics_weights(arch, seed=seed, dtype=torch.float32)
    out["weights_checksum"] = np.array(weights_checksum(sd32))
    out["meta"] = np.array([seed, B, S, N, min_len])
    batch = synth_vqa_batch(arch, B, S, N, seed=seed, min_len=min_len, dtype=torch.float32)
    for k, v in batch.items():
        out["in_" + k] = np_(v)
    g = torch.Generator().manual_seed(seed + 1)
    icv_full = torch.randn(1, arch.num_layers, arch.hidden_size, generator=g) * 0.05
    out["icv_full"] = np_(icv_full)
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        model = hf_model(arch, sd32, dt)
        iface = Interface(model, arch.pad_token_id)
        kw = dict(batch)
        kw["pixel_values"] = batch["pixel_values"].to(dt)
        with torch.no_grad():
            base = model(**kw)
        out[f"{dt_name}_logits_off"] = np_(base.logits)
        out[f"{dt_name}_image_states"] = np_(base.image_hidden_states)
        for hs_name, layers in hook_sets.items():
            lay = layers
            n_h = arch.num_layers if layers == -1 else len(layers)
            icv = icv_full[:, :n_h].contiguous()
            w = LearnableICVInterventionLMM(iface, True, lay, "model.model.layers.<LAYER_NUM>", arch.num_layers)
            raw, handles = [], []
            for blk in model.model.layers:                # registered BEFORE TraceDict -> sees raw output
                handles.append(blk.register_forward_hook(lambda m, i, o, raw=raw: raw.append(o.detach().clone())))
            edited = []
            with torch.no_grad():
                w.toggle_intervention(True)
                # second hook registered after the with-block installs TraceDict is not possible from
                # outside; read the edited value as the *input* of the next module instead.
                pre = [blk.register_forward_pre_hook(lambda m, a, k, e=edited: e.append((a[0] if a else k["hidden_states"]).detach().clone()), with_kwargs=True)
                       for blk in list(model.model.layers[1:])]
                pre.append(model.model.norm.register_forward_pre_hook(lambda m, a, e=edited: e.append(a[0].detach().clone())))
                res = w(icv=icv, **kw)
                for h_ in handles + pre:
                    h_.remove()
            out[f"{dt_name}_{hs_name}_logits"] = np_(res.logits)
            out[f"{dt_name}_{hs_name}_raw"] = np.stack([np_(t) for t in raw])
            # `edited[i]` is the input of block i+1; when a gated x-attn block precedes block i+1 the
            # input has already passed through it, so keep only the final (pre-norm) one plus those
            # of blocks not preceded by an x-attn block.
            keep = [i for i in range(arch.num_layers - 1) if (i + 1) % arch.cross_layer_interval != 0]
            out[f"{dt_name}_{hs_name}_edited_idx"] = np.array(keep + [arch.num_layers - 1])
            out[f"{dt_name}_{hs_name}_edited"] = np.stack([np_(edited[i]) for i in keep] + [np_(edited[-1])])
            out[f"{dt_name}_{hs_name}_edited_is_f32"] = np.array(edited[-1].dtype == torch.float32)
            w.toggle_intervention(False)
            with torch.no_grad():
                off = w(icv=icv, **kw)
            assert torch.equal(off.logits, base.logits)
    np.savez_compressed(OUT / f"{tag}.npz", **out)


def g3_idefics():
    _run_idefics(IDEFICS_TINY, "g3_idefics_tiny", 11, B=2, S=14, N=2, min_len=11, hook_sets={"all": -1, "sub": [1, 3]})
    _run_idefics(IDEFICS_MID, "g3_idefics_mid", 12, B=2, S=24, N=3, min_len=18, hook_sets={"all": -1})


def hf_idefics2(arch, sd, dtype):
    from transformers import Idefics2Config, Idefics2ForConditionalGeneration
    cfg = Idefics2Config(
        vision_config=dict(hidden_size=arch.v_hidden, intermediate_size=arch.v_inter, num_hidden_layers=arch.v_layers,
                           num_attention_heads=arch.v_heads, image_size=arch.v_image, patch_size=arch.v_patch,
                           hidden_act=arch.v_act, layer_norm_eps=arch.v_ln_eps),
        perceiver_config=dict(hidden_size=arch.hidden_size, resampler_n_latents=arch.r_latents, resampler_depth=arch.r_depth,
                              resampler_n_heads=arch.r_heads, resampler_head_dim=arch.r_head_dim,
                              num_key_value_heads=arch.r_kv_heads, hidden_act="silu", rms_norm_eps=arch.rms_eps),
        text_config=dict(model_type="mistral", vocab_size=arch.vocab_size, hidden_size=arch.hidden_size,
                         intermediate_size=arch.intermediate_size, num_hidden_layers=arch.num_layers,
                         num_attention_heads=arch.num_heads, num_key_value_heads=arch.num_kv_heads, rms_norm_eps=arch.rms_eps,
                         max_position_embeddings=4096, sliding_window=4096, pad_token_id=arch.pad_token_id,
                         rope_parameters=dict(rope_type="default", rope_theta=arch.rope_base)),
        image_token_id=arch.image_token_id, attn_implementation="eager")
    m = Idefics2ForConditionalGeneration(cfg)
    missing, unexpected = m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    m = m.to(dtype).eval()
    # `.to(bf16)` also rounds the rotary inv_freq BUFFER, which a checkpoint loaded with torch_dtype=bf16 never does
    # (the buffer is created in fp32 at init): restore it so the fixture reflects real usage
    rot = m.model.text_model.rotary_emb
    hd = arch.hidden_size // arch.num_heads
    rot.inv_freq = 1.0 / (arch.rope_base ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    rot.original_inv_freq = rot.inv_freq.clone()
    return m


def g4_idefics2():
    """Idefics2 tiny/mid: ragged NaViT images + one padding image, GQA, hook on `.mlp` (ref:config/lmm/idefics2-8B-base.yaml:8)
    through the reference wrapper; bf16 runs under autocast (the only way HF runs with the fp32-promoted stream)."""
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    fmt = "model.model.text_model.layers.<LAYER_NUM>.mlp"
    for tag, arch, seed, (B, S, N, ih, iw, mn) in (("g4_idefics2_tiny", IDEFICS2_TINY, 41, (2, 20, 2, 56, 42, 16)),
                                                    ("g4_idefics2_mid", IDEFICS2_MID, 42, (2, 40, 2, 84, 70, 30))):
        out = {}
        sd32 = synth_idefics2_weights(arch, seed=seed, dtype=torch.float32)
        out["weights_checksum"] = np.array(weights_checksum(sd32))
        out["meta"] = np.array([seed, B, S, N, ih, iw, mn])
        batch = synth_vqa_batch_idefics2(arch, B, S, N, ih, iw, seed=seed, min_len=mn, dtype=torch.float32,
                                         drop_last_image_of_row0=True)
        for k, v in batch.items():
            out["in_" + k] = np_(v)
        g = torch.Generator().manual_seed(seed + 1)
        icv = torch.randn(1, arch.num_layers, arch.hidden_size, generator=g) * 0.05
        out["icv_full"] = np_(icv)
        for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            model = hf_idefics2(arch, sd32, dt)
            iface = Interface(model, arch.pad_token_id)
            kw = dict(batch)
            kw["pixel_values"] = batch["pixel_values"].to(dt)
            ctx = torch.autocast("cpu", dtype=torch.bfloat16) if dt == torch.bfloat16 else contextlib.nullcontext()
            with torch.no_grad(), ctx:
                base = model(**kw)
                out[f"{dt_name}_logits_off"] = np_(base.logits)
                out[f"{dt_name}_image_hidden_states"] = np_(base.image_hidden_states)
                w = LearnableICVInterventionLMM(iface, True, -1, fmt, arch.num_layers)
                raw, outs, handles = [], [], []
                for blk in model.model.text_model.layers:
                    handles.append(blk.mlp.register_forward_hook(lambda m, i, o, raw=raw: raw.append(o.detach().clone())))
                    handles.append(blk.register_forward_hook(lambda m, i, o, outs=outs: outs.append(o.detach().clone())))
                res = w(icv=icv, **kw)
                for h_ in handles:
                    h_.remove()
                out[f"{dt_name}_all_logits"] = np_(res.logits)
                out[f"{dt_name}_all_mlp_raw"] = np.stack([np_(t) for t in raw])
                out[f"{dt_name}_all_layer_out"] = np.stack([np_(t) for t in outs])
                out[f"{dt_name}_layer_out_is_f32"] = np.array(outs[-1].dtype == torch.float32)
                w.toggle_intervention(False)
                off = w(icv=icv, **kw)
                assert torch.equal(off.logits, base.logits)
        np.savez_compressed(OUT / f"{tag}.npz", **out)


def g5_generate():
    """Hooked beam-search generate ids (beams=3, 5 new tokens, length_penalty 0 —
    ref:config/inference.yaml:26-30) + greedy, additional_vocab_size=0 (HF 5.15 beam search
    breaks with additional vocab)."""
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    arch = IDEFICS_TINY.with_(additional_vocab_size=0)
    seed = 21
    out = {}
    sd32 = synth_idefics_weights(arch, seed=seed, dtype=torch.float32)
    out["weights_checksum"] = np.array(weights_checksum(sd32))
    # N(0,0.02) weights give near-uniform logits and a degenerate "same token forever" decode; scale the
    # embedding and the head so token identity matters and beam != greedy (recorded for the tests).
    out["embed_scale"], out["head_scale"] = np.array(25.0), np.array(10.0)
    sd32["model.embed_tokens.weight"] *= 25.0
    sd32["lm_head.weight"] *= 10.0
    for pad_side in ("left", "right"):
        batch = synth_vqa_batch(arch, 3, 12, 1, seed=seed, min_len=9, dtype=torch.float32, padding_side=pad_side)
        if pad_side == "right":      # generate needs the prompt to end in real tokens: use full rows
            batch = synth_vqa_batch(arch, 3, 12, 1, seed=seed, min_len=12, dtype=torch.float32)
        for k, v in batch.items():
            out[f"{pad_side}_in_{k}"] = np_(v)
        g = torch.Generator().manual_seed(seed + 1)
        icv = torch.randn(1, arch.num_layers, arch.hidden_size, generator=g) * 0.2
        out["icv"] = np_(icv)
        for dt_name, dt in (("f32", torch.float32),):
            model = hf_model(arch, sd32, dt)
            model.generation_config.pad_token_id = arch.pad_token_id
            iface = Interface(model, arch.pad_token_id)
            w = LearnableICVInterventionLMM(iface, True, -1, "model.model.layers.<LAYER_NUM>", arch.num_layers)
            kw = dict(batch)
            with torch.inference_mode():
                beam = w.generate(icv=icv, **kw, max_new_tokens=5, num_beams=3, length_penalty=0.0,
                                  min_new_tokens=0, do_sample=False)
                greedy = w.generate(icv=icv, **kw, max_new_tokens=5, num_beams=1, do_sample=False)
                w.toggle_intervention(False)
                greedy_off = w.generate(icv=icv, **kw, max_new_tokens=5, num_beams=1, do_sample=False)
            out[f"{pad_side}_{dt_name}_beam_ids"] = beam.numpy()
            out[f"{pad_side}_{dt_name}_greedy_ids"] = greedy.numpy()
            out[f"{pad_side}_{dt_name}_greedy_off_ids"] = greedy_off.numpy()
    np.savez_compressed(OUT / "g5_generate.npz", **out)


def g8_generate_idefics2():
    """Hooked beam / greedy generate ids for Idefics2 (hook on `.mlp`), fp32, left- and right-padded prompts.
    Embedding and head scaled like g5 so that token identity matters."""
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    arch = IDEFICS2_TINY
    seed = 81
    out = {}
    sd32 = synth_idefics2_weights(arch, seed=seed, dtype=torch.float32)
    out["weights_checksum"] = np.array(weights_checksum(sd32))
    out["embed_scale"], out["head_scale"] = np.array(25.0), np.array(10.0)
    sd32["model.text_model.embed_tokens.weight"] *= 25.0
    sd32["lm_head.weight"] *= 10.0
    # the edit preserves the norm of the MLP branch: give that branch a norm comparable to the stream's so the hook can
    # change tokens (otherwise hooked == unhooked ids and the fixture would not exercise the path)
    out["down_scale"] = np.array(40.0)
    for l in range(arch.num_layers):
        sd32[f"model.text_model.layers.{l}.mlp.down_proj.weight"] *= 40.0
    fmt = "model.model.text_model.layers.<LAYER_NUM>.mlp"
    g = torch.Generator().manual_seed(seed + 1)
    icv = torch.randn(1, arch.num_layers, arch.hidden_size, generator=g) * 0.2
    out["icv"] = np_(icv)
    for pad_side in ("left", "right"):
        mn = 15 if pad_side == "left" else 20                 # right padding: full rows (the prompt must end in real tokens)
        batch = synth_vqa_batch_idefics2(arch, 3, 20, 2, 56, 42, seed=seed, min_len=mn, dtype=torch.float32, padding_side=pad_side)
        for k, v in batch.items():
            out[f"{pad_side}_in_{k}"] = np_(v)
        model = hf_idefics2(arch, sd32, torch.float32)
        model.generation_config.pad_token_id = arch.pad_token_id
        model.generation_config.eos_token_id = arch.eos_token_id
        iface = Interface(model, arch.pad_token_id)
        w = LearnableICVInterventionLMM(iface, True, -1, fmt, arch.num_layers)
        with torch.inference_mode():
            beam = w.generate(icv=icv, **batch, max_new_tokens=5, num_beams=3, length_penalty=0.0, min_new_tokens=0, do_sample=False)
            greedy = w.generate(icv=icv, **batch, max_new_tokens=5, num_beams=1, do_sample=False)
            w.toggle_intervention(False)
            greedy_off = w.generate(icv=icv, **batch, max_new_tokens=5, num_beams=1, do_sample=False)
        out[f"{pad_side}_f32_beam_ids"] = beam.numpy()
        out[f"{pad_side}_f32_greedy_ids"] = greedy.numpy()
        out[f"{pad_side}_f32_greedy_off_ids"] = greedy_off.numpy()
    np.savez_compressed(OUT / "g8_generate_idefics2.npz", **out)


JITTER_TRIALS = 24


def _gen_with_margins(w, icv, batch, num_beams, prompt_len, n_rows, seed0=1000):
    """Reference generate (through the reference's wrapper, its own call shape: ref:inference.py:313,
    ref:config/inference.yaml:26-30) returning the ids plus, per row, how ROBUST that decode is to bf16-level noise:
    the same reference call is repeated JITTER_TRIALS times with every next-token score moved by -1, 0 or +1 bf16 ulp of its own
    magnitude at random (a transformers LogitsProcessor) and the fraction of trials that reproduce the row's ids is stored.
    Two bf16 implementations of the same model differ by one ulp on a few logits per step (measured: the native engine vs this
    reference, max 1 ulp), so a row with stability 1.0 — no candidate comparison of the search sits within two ulp — must be
    reproduced exactly; rows below 1.0 are the reference's own near-ties (bf16 logits tie EXACTLY on several rows)."""
    from transformers import LogitsProcessor, LogitsProcessorList

    class Jitter(LogitsProcessor):
        def __init__(self, seed):
            self.g = torch.Generator().manual_seed(seed)

        def __call__(self, input_ids, scores):
            x = scores.float()
            ulp = torch.exp2(torch.floor(torch.log2(x.abs().clamp_min(1e-20))) - 7)
            step = torch.randint(-1, 2, x.shape, generator=self.g).float()
            return (x + step * ulp).to(scores.dtype)

    plain = dict(max_new_tokens=5, do_sample=False)
    if num_beams > 1:
        plain.update(num_beams=num_beams, length_penalty=0.0, min_new_tokens=0)
    with torch.inference_mode():
        ids = w.generate(icv=icv, **batch, **plain)
        hits = torch.zeros(n_rows)
        for t in range(JITTER_TRIALS):
            alt = w.generate(icv=icv, **batch, **plain, logits_processor=LogitsProcessorList([Jitter(seed0 + t)]))
            n = min(alt.shape[1], ids.shape[1])
            hits += ((alt[:, :n] == ids[:, :n]).all(dim=1) & (alt.shape[1] == ids.shape[1])).float()
    return ids, hits / JITTER_TRIALS


def g11_generate_bf16():
    """Hooked generate ids in the reference's own bf16 regime (ref:inference.py:300-321, ref:config/inference.yaml:26-30):
    Idefics with bf16 weights driven by the reference wrapper, 16 prompts per padding side, beam search (3 beams, 5 new tokens,
    length_penalty 0) and greedy, with and without the intervention.  Per-row stabilities (see _gen_with_margins) tell the tests
    which rows are decided by more than bf16 noise."""
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    arch = IDEFICS_TINY.with_(additional_vocab_size=0)
    seed, B = 121, 16
    out = {}
    sd32 = synth_idefics_weights(arch, seed=seed, dtype=torch.float32)
    out["weights_checksum"] = np.array(weights_checksum(sd32))
    out["embed_scale"], out["head_scale"] = np.array(25.0), np.array(10.0)
    sd32["model.embed_tokens.weight"] *= 25.0
    sd32["lm_head.weight"] *= 10.0
    g = torch.Generator().manual_seed(seed + 1)
    icv = torch.randn(1, arch.num_layers, arch.hidden_size, generator=g) * 0.2
    out["icv"] = np_(icv)
    model = hf_model(arch, sd32, torch.bfloat16)
    model.generation_config.pad_token_id = arch.pad_token_id
    iface = Interface(model, arch.pad_token_id)
    for pad_side in ("left", "right"):
        mn = 9 if pad_side == "left" else 12                  # right padding: full rows (the prompt must end in real tokens)
        batch = synth_vqa_batch(arch, B, 12, 1, seed=seed + (0 if pad_side == "left" else 7), min_len=mn, dtype=torch.float32,
                                padding_side=pad_side)
        for k, v in batch.items():
            out[f"{pad_side}_in_{k}"] = np_(v)
        kw = dict(batch)
        kw["pixel_values"] = batch["pixel_values"].to(torch.bfloat16)
        w = LearnableICVInterventionLMM(iface, True, -1, "model.model.layers.<LAYER_NUM>", arch.num_layers)
        for tag, beams, on in (("beam", 3, True), ("greedy", 1, True), ("greedy_off", 1, False)):
            w.toggle_intervention(on)
            ids, mg = _gen_with_margins(w, icv, kw, beams, 12, B)
            out[f"{pad_side}_bf16_{tag}_ids"] = ids.numpy()
            out[f"{pad_side}_bf16_{tag}_stability"] = mg.numpy()
    np.savez_compressed(OUT / "g11_generate_bf16.npz", **out)


def g12_generate_idefics2_bf16():
    """As g11 for Idefics2: bf16 weights under torch.autocast (the regime the reference needs for it), hook on
    every text layer's `.mlp`, 16 prompts per padding side with two ragged images each."""
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    arch = IDEFICS2_TINY
    seed, B = 181, 16
    out = {}
    sd32 = synth_idefics2_weights(arch, seed=seed, dtype=torch.float32)
    out["weights_checksum"] = np.array(weights_checksum(sd32))
    out["embed_scale"], out["head_scale"], out["down_scale"] = np.array(25.0), np.array(10.0), np.array(40.0)
    sd32["model.text_model.embed_tokens.weight"] *= 25.0
    sd32["lm_head.weight"] *= 10.0
    for l in range(arch.num_layers):
        sd32[f"model.text_model.layers.{l}.mlp.down_proj.weight"] *= 40.0
    fmt = "model.model.text_model.layers.<LAYER_NUM>.mlp"
    g = torch.Generator().manual_seed(seed + 1)
    icv = torch.randn(1, arch.num_layers, arch.hidden_size, generator=g) * 0.2
    out["icv"] = np_(icv)
    model = hf_idefics2(arch, sd32, torch.bfloat16)
    model.generation_config.pad_token_id = arch.pad_token_id
    model.generation_config.eos_token_id = arch.eos_token_id
    iface = Interface(model, arch.pad_token_id)
    for pad_side in ("left", "right"):
        mn = 15 if pad_side == "left" else 20
        batch = synth_vqa_batch_idefics2(arch, B, 20, 2, 56, 42, seed=seed + (0 if pad_side == "left" else 7), min_len=mn,
                                         dtype=torch.float32, padding_side=pad_side)
        for k, v in batch.items():
            out[f"{pad_side}_in_{k}"] = np_(v)
        kw = dict(batch)
        kw["pixel_values"] = batch["pixel_values"].to(torch.bfloat16)
        w = LearnableICVInterventionLMM(iface, True, -1, fmt, arch.num_layers)
        for tag, beams, on in (("beam", 3, True), ("greedy", 1, True), ("greedy_off", 1, False)):
            w.toggle_intervention(on)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                ids, mg = _gen_with_margins(w, icv, kw, beams, 20, B)
            out[f"{pad_side}_bf16_{tag}_ids"] = ids.numpy()
            out[f"{pad_side}_bf16_{tag}_stability"] = mg.numpy()
    np.savez_compressed(OUT / "g12_generate_idefics2_bf16.npz", **out)


def _select_stable(run_study, n_pool: int, n_keep: int, trials: int):
    """Rows of a candidate pool whose decode the jitter study reproduces in EVERY trial of EVERY mode (beam, greedy, hooks
    off): `run_study(rows or None, seed0, trials) -> {tag: (ids, stability)}`.  The pool is studied once, `n_keep` all-stable
    rows are kept, and the study is repeated on the kept rows alone with fresh jitter seeds (a row's arithmetic must not
    depend on its batch neighbours, but the check costs nothing); a row that fails the second study is replaced."""
    first = run_study(None, 1000, trials)
    stable = torch.ones(n_pool, dtype=torch.bool)
    for _, st in first.values():
        stable &= st >= 1.0
    cand = stable.nonzero().flatten().tolist()
    assert len(cand) >= n_keep, f"only {len(cand)} of {n_pool} candidate prompts are stable in all modes"
    keep, spare = cand[:n_keep], cand[n_keep:]
    for _ in range(8):
        second = run_study(keep, 5000, 2 * trials)
        bad = torch.zeros(len(keep), dtype=torch.bool)
        for _, st in second.values():
            bad |= st < 1.0
        if not bool(bad.any()):
            return keep, second
        for i in bad.nonzero().flatten().tolist():
            keep[i] = spare.pop(0)
    raise RuntimeError("could not find a batch that is stable in every mode")


def _gen_with_margins_seeded(w, icv, batch, num_beams, n_rows, seed0, trials):
    global JITTER_TRIALS
    old = JITTER_TRIALS
    JITTER_TRIALS = trials
    try:
        return _gen_with_margins(w, icv, batch, num_beams, 0, n_rows, seed0=seed0)
    finally:
        JITTER_TRIALS = old


def g15_generate_bf16_stable():
    """g11's set-up (the reference wrapper driving HF generate on bf16 Idefics weights, ref:inference.py:300-321,
    ref:config/inference.yaml:26-30) on WELL-CONDITIONED prompts: from 96 candidate prompts per padding side the 16 are kept
    whose decode survives every one of 24 (+48 on the kept batch) re-runs with all scores moved by -1 / 0 / +1 bf16 ulp, in
    beam, greedy and hooks-off mode alike.  No candidate comparison of those searches sits within bf16 noise, so another bf16
    implementation of the same model must reproduce EVERY row: the fixture for the north-star's "token ids bit-exact"."""
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    arch = IDEFICS_TINY.with_(additional_vocab_size=0)
    seed, POOL, B = 151, 96, 16
    out = {}
    sd32 = synth_idefics_weights(arch, seed=seed, dtype=torch.float32)
    out["weights_checksum"] = np.array(weights_checksum(sd32))
    out["embed_scale"], out["head_scale"] = np.array(25.0), np.array(10.0)
    out["weights_seed"] = np.array(seed)
    sd32["model.embed_tokens.weight"] *= 25.0
    sd32["lm_head.weight"] *= 10.0
    g = torch.Generator().manual_seed(seed + 1)
    icv = torch.randn(1, arch.num_layers, arch.hidden_size, generator=g) * 0.2
    out["icv"] = np_(icv)
    model = hf_model(arch, sd32, torch.bfloat16)
    model.generation_config.pad_token_id = arch.pad_token_id
    iface = Interface(model, arch.pad_token_id)
    for pad_side in ("left", "right"):
        mn = 9 if pad_side == "left" else 12
        pool = synth_vqa_batch(arch, POOL, 12, 1, seed=seed + (0 if pad_side == "left" else 7), min_len=mn, dtype=torch.float32,
                               padding_side=pad_side)
        w = LearnableICVInterventionLMM(iface, True, -1, "model.model.layers.<LAYER_NUM>", arch.num_layers)

        def study(rows, seed0, trials):
            b = pool if rows is None else {k: v[rows] for k, v in pool.items()}
            kw = dict(b)
            kw["pixel_values"] = b["pixel_values"].to(torch.bfloat16)
            res = {}
            for tag, beams, on in (("beam", 3, True), ("greedy", 1, True), ("greedy_off", 1, False)):
                w.toggle_intervention(on)
                res[tag] = _gen_with_margins_seeded(w, icv, kw, beams, kw["input_ids"].shape[0], seed0, trials)
            return res

        keep, res = _select_stable(study, POOL, B, 24)
        for k, v in pool.items():
            out[f"{pad_side}_in_{k}"] = np_(v[keep])
        for tag, (ids, st) in res.items():
            assert bool((st >= 1.0).all())
            out[f"{pad_side}_bf16_{tag}_ids"] = ids.numpy()
            out[f"{pad_side}_bf16_{tag}_stability"] = st.numpy()
    np.savez_compressed(OUT / "g15_generate_bf16_stable.npz", **out)


def g16_generate_idefics2_bf16_stable():
    """g12's set-up (Idefics2, bf16 under autocast, hook on every `.mlp`) on well-conditioned prompts chosen as in g15."""
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    arch = IDEFICS2_TINY
    seed, POOL, B = 191, 64, 16
    out = {}
    sd32 = synth_idefics2_weights(arch, seed=seed, dtype=torch.float32)
    out["weights_checksum"] = np.array(weights_checksum(sd32))
    out["embed_scale"], out["head_scale"], out["down_scale"] = np.array(25.0), np.array(10.0), np.array(40.0)
    out["weights_seed"] = np.array(seed)
    sd32["model.text_model.embed_tokens.weight"] *= 25.0
    sd32["lm_head.weight"] *= 10.0
    for l in range(arch.num_layers):
        sd32[f"model.text_model.layers.{l}.mlp.down_proj.weight"] *= 40.0
    fmt = "model.model.text_model.layers.<LAYER_NUM>.mlp"
    g = torch.Generator().manual_seed(seed + 1)
    icv = torch.randn(1, arch.num_layers, arch.hidden_size, generator=g) * 0.2
    out["icv"] = np_(icv)
    model = hf_idefics2(arch, sd32, torch.bfloat16)
    model.generation_config.pad_token_id = arch.pad_token_id
    model.generation_config.eos_token_id = arch.eos_token_id
    iface = Interface(model, arch.pad_token_id)
    for pad_side in ("left", "right"):
        mn = 15 if pad_side == "left" else 20
        pool = synth_vqa_batch_idefics2(arch, POOL, 20, 2, 56, 42, seed=seed + (0 if pad_side == "left" else 7), min_len=mn,
                                        dtype=torch.float32, padding_side=pad_side)
        w = LearnableICVInterventionLMM(iface, True, -1, fmt, arch.num_layers)

        def study(rows, seed0, trials):
            b = pool if rows is None else {k: v[rows] for k, v in pool.items()}
            kw = dict(b)
            kw["pixel_values"] = b["pixel_values"].to(torch.bfloat16)
            res = {}
            for tag, beams, on in (("beam", 3, True), ("greedy", 1, True), ("greedy_off", 1, False)):
                w.toggle_intervention(on)
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    res[tag] = _gen_with_margins_seeded(w, icv, kw, beams, kw["input_ids"].shape[0], seed0, trials)
            return res

        keep, res = _select_stable(study, POOL, B, 24)
        for k, v in pool.items():
            out[f"{pad_side}_in_{k}"] = np_(v[keep])
        for tag, (ids, st) in res.items():
            assert bool((st >= 1.0).all())
            out[f"{pad_side}_bf16_{tag}_ids"] = ids.numpy()
            out[f"{pad_side}_bf16_{tag}_stability"] = st.numpy()
    np.savez_compressed(OUT / "g16_generate_idefics2_bf16_stable.npz", **out)


def g17_image_preprocess():
    """f2, second half: what `processor.prepare_input` does to an image before the model sees it (ref:icv_src/icv_datamodule.py:80-124
    through lmm_icl_interface -> the HF image processors): uint8 HWC -> rescale by 1/255 -> (x - mean) / std -> CHW float32 [-> the
    model's bf16].  Idefics: HF's own `IdeficsImageProcessorPil` (resizing off: the dataset side resizes with PIL, a CPU step that
    stays there) on seeded uint8 images.  Idefics2: `Idefics2ImageProcessorPil` cannot be constructed here (it asks for torchvision,
    which is not installed: an ordinary ImportError), so the fixture uses the two numpy functions it calls,
    `transformers.image_transforms.rescale` and `.normalize`, with its mean / std (0.5) and its padding rule (zeros to the right and
    below up to the batch maximum, pixel_attention_mask = 1 on real pixels) restated from hf:idefics2/image_processing_pil_idefics2.py."""
    from transformers.image_transforms import normalize, rescale
    from transformers.models.idefics.image_processing_pil_idefics import IdeficsImageProcessorPil
    rng = np.random.default_rng(1717)
    out = {}
    proc = IdeficsImageProcessorPil(do_resize=False)
    imgs = [rng.integers(0, 256, (56, 56, 3)).astype(np.uint8) for _ in range(6)]
    imgs[0][:] = np.arange(256, dtype=np.uint8).repeat(37)[: 56 * 56 * 3].reshape(56, 56, 3)      # every byte value in every channel
    imgs[1][..., 0] = np.arange(56 * 56).reshape(56, 56) % 256
    imgs[1][..., 1] = (np.arange(56 * 56).reshape(56, 56) * 7 + 3) % 256
    imgs[1][..., 2] = 255 - imgs[1][..., 0]
    pv = proc.preprocess(imgs, return_tensors="pt")
    out["idefics_u8"] = np.stack(imgs)
    out["idefics_f32"] = pv.numpy().astype(np.float32)
    out["idefics_mean"] = np.array(proc.image_mean, dtype=np.float64)
    out["idefics_std"] = np.array(proc.image_std, dtype=np.float64)
    out["idefics_rescale"] = np.array(proc.rescale_factor, dtype=np.float64)
    # Idefics2: ragged images, (B = 2, N = 2) with one missing image (an all-zero padding image with an all-zero mask)
    mean2 = std2 = [0.5, 0.5, 0.5]
    sizes = [[(56, 42), (28, 70)], [(42, 42), None]]
    Hm, Wm = 56, 70
    u8 = np.zeros((2, 2, Hm, Wm, 3), dtype=np.uint8)
    hw = np.zeros((2, 2, 2), dtype=np.int32)
    exp = np.zeros((2, 2, 3, Hm, Wm), dtype=np.float32)
    mask = np.zeros((2, 2, Hm, Wm), dtype=np.int64)
    for b in range(2):
        for n in range(2):
            if sizes[b][n] is None:
                continue
            h, w = sizes[b][n]
            im = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            u8[b, n, :h, :w] = im
            hw[b, n] = (h, w)
            x = normalize(rescale(im, 1 / 255), mean2, std2)                   # (h, w, 3) float32
            exp[b, n, :, :h, :w] = np.transpose(x, (2, 0, 1))
            mask[b, n, :h, :w] = 1
    out["idefics2_u8"], out["idefics2_hw"] = u8, hw
    out["idefics2_f32"], out["idefics2_mask"] = exp, mask
    np.savez_compressed(OUT / "g17_image_preprocess.npz", **out)


def g13_frontend():
    """Processor-side integer rules, produced by the HF code itself:
    (a) Idefics image_attention_mask: transformers' image_attention_mask_for_packed_input_ids_pt + incremental_to_binary_attention_mask
        on random id rows with <image> / end-of-document tokens (incl. more images than mask columns, leading text, EOD runs);
    (b) Idefics2: the patch_attention_mask Idefics2Model.forward derives, which images it drops as padding, and the NaViT
        position ids Idefics2VisionEmbeddings feeds its position table — captured with forward hooks on a tiny HF model (bf16), and
        on a stand-alone embeddings module at the full 980 x 980 / 70 x 70 grid."""
    import types as _t
    from transformers.models.idefics.processing_idefics import (image_attention_mask_for_packed_input_ids_pt,
                                                                incremental_to_binary_attention_mask)
    out = {}
    g = torch.Generator().manual_seed(1301)
    IMG, EOD = 11, 2
    tok = _t.SimpleNamespace(convert_tokens_to_ids=lambda t: IMG, eos_token_id=EOD)
    for tag, (B, S, hi, n_cls) in dict(a=(6, 150, 14, 5), b=(3, 64, 40, 3), c=(2, 1, 12, 2), d=(4, 333, 13, 33)).items():
        ids = torch.randint(0, hi, (B, S), generator=g)
        ids[0, : min(S, 9)] = 5                                  # a row that starts with text only
        if S > 40:
            ids[1, 20:30] = EOD                                   # a run of end-of-document tokens
        inc, _ = image_attention_mask_for_packed_input_ids_pt(ids.clone(), tok)
        mask = incremental_to_binary_attention_mask(inc, "pt", num_classes=n_cls)
        out[f"m_{tag}_ids"], out[f"m_{tag}_mask"], out[f"m_{tag}_cfg"] = ids.numpy(), mask.numpy().astype(np.int8), np.array([IMG, EOD, n_cls])
    # (b) tiny HF Idefics2 with ragged images and one all-zero padding image
    for tag, arch, hw in (("tiny", IDEFICS2_TINY, (56, 42)), ("mid", IDEFICS2_MID, (84, 70))):
        sd = synth_idefics2_weights(arch, seed=1302, dtype=torch.float32)
        model = hf_idefics2(arch, sd, torch.bfloat16)
        batch = synth_vqa_batch_idefics2(arch, 3, 40 if tag == "tiny" else 60, 2, hw[0], hw[1], seed=1303, min_len=30 if tag == "tiny" else 50,
                                         dtype=torch.float32, drop_last_image_of_row0=True)
        cap = {}
        emb = model.model.vision_model.embeddings
        def grab_pos(m, a):
            cap["pos"] = a[0].clone()

        def grab_mask(m, a, kw):
            cap["pam"], cap["n"] = kw["patch_attention_mask"].clone(), kw["pixel_values"].shape[0]
        h1 = emb.position_embedding.register_forward_pre_hook(grab_pos)
        h2 = model.model.vision_model.register_forward_pre_hook(grab_mask, with_kwargs=True)
        with torch.inference_mode(), torch.autocast("cpu", dtype=torch.bfloat16):
            model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], pixel_values=batch["pixel_values"].to(torch.bfloat16),
                  pixel_attention_mask=batch["pixel_attention_mask"])
        h1.remove(); h2.remove()
        out[f"v_{tag}_pixel_values"] = np_(batch["pixel_values"])
        out[f"v_{tag}_pixel_attention_mask"] = batch["pixel_attention_mask"].numpy()
        out[f"v_{tag}_patch_mask"] = cap["pam"].numpy()
        out[f"v_{tag}_position_ids"] = cap["pos"].numpy()
        out[f"v_{tag}_n_real"] = np.array(cap["n"])
    # full-size grid: the embeddings module alone (hidden 16), 980 x 980 canvas, three ragged images
    from transformers.models.idefics2.configuration_idefics2 import Idefics2VisionConfig
    from transformers.models.idefics2.modeling_idefics2 import Idefics2VisionEmbeddings
    vc = Idefics2VisionConfig(hidden_size=16, image_size=980, patch_size=14, num_hidden_layers=1, num_attention_heads=1, intermediate_size=16)
    emb = Idefics2VisionEmbeddings(vc).to(torch.bfloat16).eval()
    sizes = [(980, 980), (378, 504), (14, 966), (700, 28)]
    pam = torch.zeros(len(sizes), 980, 980, dtype=torch.bool)
    for i, (hh, ww) in enumerate(sizes):
        pam[i, :hh, :ww] = True
    sub = pam.unfold(1, 14, 14).unfold(2, 14, 14)
    patch_mask = (sub.sum(dim=(-1, -2)) == 14 * 14).bool()
    cap = {}
    def grab_full(m, a):
        cap["pos"] = a[0].clone()
    h1 = emb.position_embedding.register_forward_pre_hook(grab_full)
    with torch.inference_mode():
        emb(torch.zeros(len(sizes), 3, 980, 980, dtype=torch.bfloat16), patch_mask)
    h1.remove()
    out["v_full_sizes"] = np.array(sizes)
    out["v_full_position_ids"] = cap["pos"].numpy().astype(np.int16)
    np.savez_compressed(OUT / "g13_frontend.npz", **out)


def g14_vqa_metric():
    """VQA accuracy scoring: the reference's own evaluator (icv_src/metrics/vqa_metric.py imports with the
    standard library alone) on a corpus of answer strings — every key of its contraction / number-word tables, punctuation and
    digit cases — and on synthetic annotation / question / result files."""
    import contextlib as _c
    import importlib.util
    import io
    import json as _json
    import tempfile
    spec = importlib.util.spec_from_file_location("ref_vqa_metric", REF / "icv_src" / "metrics" / "vqa_metric.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    ev = ref.VQAEval(None, None)
    corpus = sorted(ev.contractions) + sorted(set(ev.contractions.values())) + list(ev.manualMap) + list(ev.articles)
    corpus += ["Yes.", "no!", "a red bus", "The  Two dogs", "1,000", "3.5", "it's 2,5 meters, ok", "U.S.A.", "dont know; maybe", "left/right",
               "x - y", "hello , world", "tennis racket?", "An apple (green)", "e.g. this", "....", "a.b.c.d.e.f.g.h.i.j.k.l.m.n.o.p.q.r.s.t.u.v.w.x.y.z.a.b.c.d.e.f.g.h.i",
               "couldntve done it", "Shes here", "lets go", "somebody'd know", "10", "ten", "None", "\tspaced\n out ", "a", "", "#1 player @home", "50% off"]
    g = torch.Generator().manual_seed(1401)
    vocab = ["yes", "no", "2", "two", "red", "a red", "the bus", "bus.", "dont", "don't", "tennis", "Tennis racket", "racket,", "1,000", "left", "right"]
    corpus += [" ".join(vocab[int(i)] for i in torch.randint(0, len(vocab), (int(torch.randint(1, 4, (1,), generator=g)),), generator=g)) for _ in range(60)]
    out = {"corpus": np.array(corpus)}
    out["punct"] = np.array([ev.processPunctuation(t.replace("\n", " ").replace("\t", " ").strip()) for t in corpus])
    out["full"] = np.array([ev.processDigitArticle(ev.processPunctuation(t.replace("\n", " ").replace("\t", " ").strip())) for t in corpus])
    gens = ["cat Question: what", "two, maybe three Answer: x", "Short answer: blue", " red\nQuestion", "a dog", "yes, it is", ""]
    out["gen_in"] = np.array(gens)
    out["gen_out"] = np.array([ref.postprocess_vqa_generation(t) for t in gens])
    # synthetic files: 40 questions, 10 human answers each, 3 question types, answer types incl. a missing one
    anns, ques, res = [], [], []
    for q in range(40):
        humans = [vocab[int(i)] for i in torch.randint(0, len(vocab), (10,), generator=g)]
        a = dict(question_id=1000 + q, image_id=q // 3, question_type=["what", "is the", "how many"][q % 3],
                 multiple_choice_answer=humans[0],
                 answers=[dict(answer=h, answer_confidence="yes", answer_id=i + 1) for i, h in enumerate(humans)])
        if q % 5:
            a["answer_type"] = ["yes/no", "number", "other"][q % 3]
        anns.append(a)
        ques.append(dict(question_id=1000 + q, image_id=q // 3, question="synthetic?"))
        res.append(dict(question_id=1000 + q, answer=vocab[int(torch.randint(0, len(vocab), (1,), generator=g))] if q % 7 else humans[2] + "!"))
    meta = dict(info={}, task_type="Open-Ended", data_type="mscoco", data_subtype="val", license={})
    files = dict(ann=dict(meta, annotations=anns), que=dict(meta, questions=ques), res=res)
    with tempfile.TemporaryDirectory() as d:
        paths = {}
        for k, v in files.items():
            paths[k] = str(Path(d) / f"{k}.json")
            _json.dump(v, open(paths[k], "w"))
        with _c.redirect_stdout(io.StringIO()):
            acc = ref.compute_vqa_accuracy(paths["res"], paths["que"], paths["ann"])
    for k, v in files.items():
        out[f"file_{k}"] = np.array(_json.dumps(v))
    out["accuracy"] = np.array(_json.dumps(acc))
    np.savez_compressed(OUT / "g14_vqa_metric.npz", **out)


def g6_loss():
    """The reference's VQAICVModule.forward (student hooked + teacher plain + KL) and its grads."""
    from icv_src.icv_encoder.global_icv_encoder import GlobalICVEncoder
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    methods = _reference_module_methods()
    arch = IDEFICS_TINY
    seed = 31
    sd32 = synth_idefics_weights(arch, seed=seed, dtype=torch.float32)
    out = {"weights_checksum": np.array(weights_checksum(sd32))}
    # teacher batch: 2 shots + query (3 images); student batch: query only (1 image).  The answer span
    # is the last `ans` real tokens of each row, identical in both (collator contract).
    B, ans = 2, 3
    tea = synth_vqa_batch(arch, B, 22, 3, seed=seed, min_len=18, dtype=torch.float32)
    stu = synth_vqa_batch(arch, B, 10, 1, seed=seed + 1, min_len=8, dtype=torch.float32)
    tl = tea["attention_mask"].sum(1)
    sl = stu["attention_mask"].sum(1)
    for b in range(B):                                   # same answer tokens at the tail of both rows
        stu["input_ids"][b, sl[b] - ans: sl[b]] = tea["input_ids"][b, tl[b] - ans: tl[b]]
    in_context_length = tl - ans
    query_x_length = sl - ans
    for name, d in (("tea", tea), ("stu", stu)):
        for k, v in d.items():
            out[f"{name}_{k}"] = np_(v)
    out["in_context_length"] = in_context_length.numpy()
    out["query_x_length"] = query_x_length.numpy()

    class Mod(torch.nn.Module):
        pass
    for k, f in methods.items():
        setattr(Mod, k, f)
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        for T in (1.0, 2.0):
            model = hf_model(arch, sd32, dt)
            iface = Interface(model, arch.pad_token_id)
            iface.requires_grad_(False)
            mod = Mod()
            mod.interface = iface
            mod.module_cfg = types.SimpleNamespace(hard_loss_weight=0.0, only_hard_loss=False, kl_eps=1e-6)
            mod.icv_model = LearnableICVInterventionLMM(iface, True, -1, "model.model.layers.<LAYER_NUM>", arch.num_layers)
            torch.manual_seed(seed)
            mod.icv_encoder = GlobalICVEncoder(arch.hidden_size, arch.num_layers, alpha_init_value=0.3, use_sigmoid=True)
            with torch.no_grad():
                mod.icv_encoder.icv.mul_(20.0)              # make the student differ visibly from zero-shot
            mod.temperature = torch.nn.Parameter(torch.tensor(T), requires_grad=False)
            q = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in stu.items()}
            t = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in tea.items()}
            loss_dict, enc_out = mod(q, t, query_x_length, in_context_length)
            loss_dict["loss"].backward()
            key = f"{dt_name}_T{int(T)}"
            out[f"{key}_kl"] = np_(loss_dict["kl_loss"])
            out[f"{key}_loss"] = np_(loss_dict["loss"])
            out[f"{key}_grad_icv"] = np_(mod.icv_encoder.icv.grad)
            out[f"{key}_grad_alpha"] = np_(mod.icv_encoder.alpha.grad)
            if key == "f32_T1":
                out["enc_icv"] = np_(mod.icv_encoder.icv)
                out["enc_alpha_param"] = np_(mod.icv_encoder.alpha)
                out["stu_mask"] = mod.get_mask(q, query_x_length).numpy()
                out["tea_mask"] = mod.get_mask(t, in_context_length).numpy()
                with torch.no_grad():
                    mod.icv_model.toggle_intervention(True)
                    icv_eff = enc_out.alpha.unsqueeze(-1) * enc_out.in_context_vector
                    lg = mod.icv_model(icv=icv_eff, **q)["logits"]
                    out["f32_student_logits"] = np_(lg)
                    # CE as the pinned transformers 4.38.2 Idefics forward computed it (pads masked by
                    # attention_mask) == HF 5.x loss with pad labels set to -100.
                    labels = q["input_ids"].masked_fill(q["attention_mask"] == 0, -100)
                    sl_, sh = lg[:, :-1].float().reshape(-1, lg.shape[-1]), labels[:, 1:].reshape(-1)
                    out["f32_student_ce"] = np_(torch.nn.functional.cross_entropy(sl_, sh, ignore_index=-100))
    np.savez_compressed(OUT / "g6_loss.npz", **out)


def g9_loss_idefics2():
    """g6 for Idefics2: the reference's VQAICVModule.forward (student hooked on every `.mlp` + teacher plain + KL) and the
    grads of icv / alpha through HF Idefics2; bf16 under autocast (the regime the reference uses for this model)."""
    from icv_src.icv_encoder.global_icv_encoder import GlobalICVEncoder
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    methods = _reference_module_methods()
    arch = IDEFICS2_TINY
    seed = 91
    sd32 = synth_idefics2_weights(arch, seed=seed, dtype=torch.float32)
    out = {"weights_checksum": np.array(weights_checksum(sd32))}
    B, ans = 2, 3
    tea = synth_vqa_batch_idefics2(arch, B, 36, 3, 56, 42, seed=seed, min_len=32, dtype=torch.float32)
    stu = synth_vqa_batch_idefics2(arch, B, 18, 1, 56, 42, seed=seed + 1, min_len=15, dtype=torch.float32)
    tl, sl = tea["attention_mask"].sum(1), stu["attention_mask"].sum(1)
    for b in range(B):
        stu["input_ids"][b, sl[b] - ans: sl[b]] = tea["input_ids"][b, tl[b] - ans: tl[b]]
    in_context_length, query_x_length = tl - ans, sl - ans
    for name, d in (("tea", tea), ("stu", stu)):
        for k, v in d.items():
            out[f"{name}_{k}"] = np_(v)
    out["in_context_length"], out["query_x_length"] = in_context_length.numpy(), query_x_length.numpy()

    class Mod(torch.nn.Module):
        pass
    for k, f in methods.items():
        setattr(Mod, k, f)
    fmt = "model.model.text_model.layers.<LAYER_NUM>.mlp"
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        for T in (1.0, 2.0):
            model = hf_idefics2(arch, sd32, dt)
            iface = Interface(model, arch.pad_token_id)
            iface.requires_grad_(False)
            mod = Mod()
            mod.interface = iface
            mod.module_cfg = types.SimpleNamespace(hard_loss_weight=0.0, only_hard_loss=False, kl_eps=1e-6)
            mod.icv_model = LearnableICVInterventionLMM(iface, True, -1, fmt, arch.num_layers)
            torch.manual_seed(seed)
            mod.icv_encoder = GlobalICVEncoder(arch.hidden_size, arch.num_layers, alpha_init_value=0.3, use_sigmoid=True)
            with torch.no_grad():
                mod.icv_encoder.icv.mul_(5.0)
            mod.temperature = torch.nn.Parameter(torch.tensor(T), requires_grad=False)
            q = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in stu.items()}
            t = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in tea.items()}
            ctx = torch.autocast("cpu", dtype=torch.bfloat16) if dt == torch.bfloat16 else contextlib.nullcontext()
            with ctx:
                loss_dict, enc_out = mod(q, t, query_x_length, in_context_length)
            loss_dict["loss"].backward()
            key = f"{dt_name}_T{int(T)}"
            out[f"{key}_kl"] = np_(loss_dict["kl_loss"])
            out[f"{key}_grad_icv"] = np_(mod.icv_encoder.icv.grad)
            out[f"{key}_grad_alpha"] = np_(mod.icv_encoder.alpha.grad)
            if key == "f32_T1":
                out["enc_icv"] = np_(mod.icv_encoder.icv)
                out["enc_alpha_param"] = np_(mod.icv_encoder.alpha)
                out["stu_mask"] = mod.get_mask(q, query_x_length).numpy()
                out["tea_mask"] = mod.get_mask(t, in_context_length).numpy()
    np.savez_compressed(OUT / "g9_loss_idefics2.npz", **out)


def g10_hard_loss():
    """The reference's objective with the CE ("hard") term: loss = kl + 0.5 * ce (ref:icv_src/icv_module.py:94-95,111-117), grads
    of icv / alpha.  additional_vocab_size = 0 (HF 5.15's labels path raises with additional vocabulary) and full-length rows
    (no padding), so the pinned 4.38.2 CE (pads masked by attention_mask) and 5.15's CE coincide."""
    from icv_src.icv_encoder.global_icv_encoder import GlobalICVEncoder
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    methods = _reference_module_methods()
    arch = IDEFICS_TINY.with_(additional_vocab_size=0)
    seed = 101
    sd32 = synth_idefics_weights(arch, seed=seed, dtype=torch.float32)
    out = {"weights_checksum": np.array(weights_checksum(sd32))}
    B, ans = 2, 3
    tea = synth_vqa_batch(arch, B, 22, 3, seed=seed, min_len=22, dtype=torch.float32)
    stu = synth_vqa_batch(arch, B, 10, 1, seed=seed + 1, min_len=10, dtype=torch.float32)
    for b in range(B):
        stu["input_ids"][b, 10 - ans:] = tea["input_ids"][b, 22 - ans:]
    in_context_length = torch.full((B,), 22 - ans)
    query_x_length = torch.full((B,), 10 - ans)
    for name, d in (("tea", tea), ("stu", stu)):
        for k, v in d.items():
            out[f"{name}_{k}"] = np_(v)
    out["in_context_length"], out["query_x_length"] = in_context_length.numpy(), query_x_length.numpy()
    out["hard_loss_weight"] = np.array(0.5)

    class Mod(torch.nn.Module):
        pass
    for k, f in methods.items():
        setattr(Mod, k, f)
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        model = hf_model(arch, sd32, dt)
        iface = Interface(model, arch.pad_token_id)
        iface.requires_grad_(False)
        mod = Mod()
        mod.interface = iface
        mod.module_cfg = types.SimpleNamespace(hard_loss_weight=0.5, only_hard_loss=False, kl_eps=1e-6)
        mod.icv_model = LearnableICVInterventionLMM(iface, True, -1, "model.model.layers.<LAYER_NUM>", arch.num_layers)
        torch.manual_seed(seed)
        mod.icv_encoder = GlobalICVEncoder(arch.hidden_size, arch.num_layers, alpha_init_value=0.3, use_sigmoid=True)
        with torch.no_grad():
            mod.icv_encoder.icv.mul_(20.0)
        mod.temperature = torch.nn.Parameter(torch.tensor(1.0), requires_grad=False)
        q = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in stu.items()}
        t = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in tea.items()}
        loss_dict, enc_out = mod(q, t, query_x_length, in_context_length)
        loss_dict["loss"].backward()
        out[f"{dt_name}_kl"] = np_(loss_dict["kl_loss"])
        out[f"{dt_name}_ce"] = np_(loss_dict["ce_loss"])
        out[f"{dt_name}_loss"] = np_(loss_dict["loss"])
        out[f"{dt_name}_grad_icv"] = np_(mod.icv_encoder.icv.grad)
        out[f"{dt_name}_grad_alpha"] = np_(mod.icv_encoder.alpha.grad)
        if dt_name == "f32":
            out["enc_icv"] = np_(mod.icv_encoder.icv)
            out["enc_alpha_param"] = np_(mod.icv_encoder.alpha)
    np.savez_compressed(OUT / "g10_hard_loss.npz", **out)


def g7_optim():
    """torch.optim.AdamW with the reference's two param groups + transformers cosine warm-up
    (ref:icv_src/icv_module.py:171-209; icv_module.yaml: alpha_lr 1e-2, icv_lr 1e-4, wd 1e-3, warm 0.1)."""
    from transformers import get_cosine_schedule_with_warmup
    g = torch.Generator().manual_seed(5)
    icv = torch.nn.Parameter(torch.randn(1, 4, 32, generator=g) * 0.01)
    alpha = torch.nn.Parameter(torch.full((1, 4), 0.2))
    opt = torch.optim.AdamW([{"params": alpha, "lr": 1e-2}, {"params": icv}], lr=1e-4, weight_decay=1e-3)
    total, warm = 20, 0.1 * 20
    sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total)
    out = {"icv0": np_(icv), "alpha0": np_(alpha), "total": np.array(total), "warm": np.array(warm)}
    grads_i, grads_a, lrs, icvs, alphas = [], [], [], [], []
    for step in range(6):
        gi = torch.randn(1, 4, 32, generator=g) * 3.0
        ga = torch.randn(1, 4, generator=g) * 3.0
        icv.grad, alpha.grad = gi.clone(), ga.clone()
        torch.nn.utils.clip_grad_norm_([alpha, icv], 1.0)
        grads_i.append(np_(gi)); grads_a.append(np_(ga))
        lrs.append([pg["lr"] for pg in opt.param_groups])
        opt.step(); sch.step()
        icvs.append(np_(icv).copy()); alphas.append(np_(alpha).copy())
    out.update(grads_icv=np.stack(grads_i), grads_alpha=np.stack(grads_a), lrs=np.array(lrs),
               icv_steps=np.stack(icvs), alpha_steps=np.stack(alphas))
    np.savez_compressed(OUT / "g7_optim.npz", **out)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    _install_shims()
    sys.path.insert(0, str(REF))
    import icv_src.icv_model.icv_intervention as _ri
    assert _ri.__file__.startswith(str(REF)), _ri.__file__
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17"]
    fns = dict(g1=g1_encoder, g2=g2_intervention, g3=g3_idefics, g4=g4_idefics2, g5=g5_generate, g6=g6_loss, g7=g7_optim, g8=g8_generate_idefics2, g9=g9_loss_idefics2, g10=g10_hard_loss,
               g11=g11_generate_bf16, g12=g12_generate_idefics2_bf16, g13=g13_frontend, g14=g14_vqa_metric,
               g15=g15_generate_bf16_stable, g16=g16_generate_idefics2_bf16_stable, g17=g17_image_preprocess)
    for w in which:
        print("generating", w, flush=True)
        fns[w]()
    for p in sorted(OUT.glob("*.npz")):
        print(f"{p.name}: {p.stat().st_size/1024:.0f} KiB")


if __name__ == "__main__":
    main()
