"""CPU checks of the drop-in surface: same names, arguments, attributes and errors as the reference
(ref:icv_src/icv_encoder/*, ref:icv_src/icv_model/icv_intervention.py, ref:icv_src/icv_module.py), pinned by the
fixtures the reference's own classes produced (g1, g2)."""
import types

import pytest
import torch

from icv_src.icv_encoder.base_icv_encoder import BaseICVEncoder, ICVEncoderOutput
from icv_src.icv_encoder.global_icv_encoder import GlobalICVEncoder
from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM

T = torch.from_numpy


def test_encoder_is_bit_identical_to_reference_under_same_seed(golden):
    z = golden("g1_encoder")
    for tag in ("a", "b"):
        H, L, a0, sig = z[f"{tag}_cfg"]
        torch.manual_seed(426)
        enc = GlobalICVEncoder(lmm_hidden_dim=int(H), lmm_layers=int(L), alpha_init_value=float(a0), use_sigmoid=bool(sig))
        out = enc()
        assert isinstance(enc, BaseICVEncoder) and isinstance(out, ICVEncoderOutput)
        assert out.in_context_feature is None
        assert torch.equal(out.in_context_vector, T(z[f"{tag}_icv"]))
        assert torch.equal(enc.alpha.detach(), T(z[f"{tag}_alpha_param"]))
        assert torch.equal(out.alpha.detach(), T(z[f"{tag}_alpha_out"]))
        assert sorted(enc.state_dict().keys()) == list(z[f"{tag}_state_keys"])
    assert not GlobalICVEncoder(8, 2, alpha_learnable=False).alpha.requires_grad


def test_intervention_wrapper_attributes_and_errors(golden):
    z = golden("g2_intervention")
    lmm = torch.nn.Linear(2, 2)
    w = LearnableICVInterventionLMM(lmm, enable_intervention=True, intervention_layer=[3, 7, 1],
                                    layer_format="model.model.layers.<LAYER_NUM>", total_layers=8)
    assert w.intervention_layer_names == list(z["names"])
    assert list(w.layer_to_icv_index.keys()) == list(z["map_keys"]) and list(w.layer_to_icv_index.values()) == list(z["map_vals"])
    assert LearnableICVInterventionLMM(lmm, True, -1, "blk.<LAYER_NUM>.mlp", 5).intervention_layer_names == list(z["names_all"])
    assert LearnableICVInterventionLMM(lmm, True, 2, "blk.<LAYER_NUM>", 5).intervention_layer_names == list(z["names_int"])
    assert w.intervention_status is True and w.lmm is lmm and w.total_layers == 8
    w.toggle_intervention(False)
    assert w.intervention_enabled is False
    with pytest.raises(ValueError) as e:
        w.toggle_intervention(1)
    assert str(e.value) == str(z["toggle_error"])
    off = LearnableICVInterventionLMM(lmm, enable_intervention=False)
    assert not hasattr(off, "intervention_layers") and not hasattr(off, "layer_to_icv_index")
    # disabled wrapper is a passthrough (also on CPU: no kernel is involved)
    x = torch.randn(3, 2)
    assert torch.equal(off(None, x), lmm(x))
    w.toggle_intervention(False)
    assert torch.equal(w(None, x), lmm(x))


def test_enabled_intervention_without_icv_or_with_unknown_layer_fails_loudly():
    lmm = torch.nn.Sequential(torch.nn.Linear(2, 2))
    w = LearnableICVInterventionLMM(lmm, True, [0], "<LAYER_NUM>", 1)
    with pytest.raises(ValueError, match="no `icv`"):
        w(None, torch.randn(1, 2))
    w2 = LearnableICVInterventionLMM(lmm, True, [5], "layers.<LAYER_NUM>", 8)
    with pytest.raises(LookupError):
        w2(torch.zeros(1, 1, 2), torch.randn(1, 2))


def test_checkpoint_dict_layout_round_trip(tmp_path):
    """icv_cpk.pth layout written by ref:train.py:97-106 and read by ref:inference.py:95-100."""
    enc = GlobalICVEncoder(16, 3, alpha_init_value=0.1, use_sigmoid=True)
    ck = {"icv_encoder.alpha": enc.alpha.detach().clone(), "icv_encoder.icv": enc.icv.detach().clone(), "use_sigmoid": True,
          "lmm_args": {"intervention_layer": -1, "layer_format": "model.model.layers.<LAYER_NUM>", "total_layers": 3}}
    torch.save(ck, tmp_path / "icv_cpk.pth")
    back = torch.load(tmp_path / "icv_cpk.pth")
    enc2 = GlobalICVEncoder(16, 3)
    enc2.load_state_dict({k.split(".", 1)[1]: v for k, v in back.items() if k.startswith("icv_encoder.")})
    assert torch.equal(enc2.icv, enc.icv) and torch.equal(enc2.alpha, enc.alpha)
    args = dict(back["lmm_args"])
    w = LearnableICVInterventionLMM(torch.nn.Identity(), True, args["intervention_layer"], args["layer_format"], args["total_layers"])
    assert w.intervention_layer_names == [f"model.model.layers.{i}" for i in range(3)]


def test_collator_contract_masks_select_the_same_answer_tokens():
    """ref:icv_src/icv_datamodule.py:73-130 contract with a whitespace tokenizer: student and teacher masks built from
    query_x_length / in_context_length (ref:icv_src/icv_module.py:136-148) pick the same answer tokens (+EOS) in both rows."""
    import types
    import torch
    from icv_src.icv_datamodule import VQAICVDataModule, collator_data

    PAD, BOS, EOS = 0, 1, 2
    vocab = {}

    class Proc:
        input_ids_field = "input_ids"
        tokenizer = types.SimpleNamespace(pad_token_id=PAD, bos_token_id=BOS, eos_token_id=EOS, padding_side="left")

        def prepare_input(self, batch_prompts, padding=True, truncation=None, add_eos_token=False, return_tensors="pt", **kw):
            rows = []
            for prompt in batch_prompts:                      # prompt = list of items; strings are text, anything else an image
                ids = [BOS]
                for item in prompt:
                    if isinstance(item, str):
                        ids += [vocab.setdefault(w, 10 + len(vocab)) for w in item.split()]
                    else:
                        ids += [5, 6, 5]                      # <fake><image><fake>
                if add_eos_token:
                    ids.append(EOS)
                rows.append(ids)
            n = max(map(len, rows))
            ids = torch.tensor([r + [PAD] * (n - len(r)) for r in rows])
            return {"input_ids": ids, "attention_mask": (ids != PAD).long()}

    img = object()
    samples = [
        dict(ice_prompt=[img, "Question: what colour ? Short answer: red", img, "Question: how many ? Short answer: two"],
             query_prompt=[img, "Question: what animal is this ? Short answer: a small dog"],
             query_x=[img, "Question: what animal is this ? Short answer:"]),
        dict(ice_prompt=[img, "Question: is it day ? Short answer: yes"],
             query_prompt=[img, "Question: where ? Short answer: beach"],
             query_x=[img, "Question: where ? Short answer:"]),
    ]
    proc = Proc()
    dm = VQAICVDataModule(types.SimpleNamespace(bs=2, num_workers=0), None, proc)
    assert proc.tokenizer.padding_side == "right"
    out = dm.collator_data(samples)
    assert set(out) == {"query_inputs", "inputs", "in_context_length", "query_x_length"}
    assert torch.equal(out["in_context_length"], collator_data(samples, proc)["in_context_length"])
    stu, tea = out["query_inputs"]["input_ids"], out["inputs"]["input_ids"]

    def mask(ids, length):
        steps = torch.arange(ids.shape[1]).unsqueeze(0).expand(ids.shape[0], -1)
        return (steps >= length.unsqueeze(1)) & (ids != PAD)

    ms, mt = mask(stu, out["query_x_length"]), mask(tea, out["in_context_length"])
    assert torch.equal(ms.sum(1), mt.sum(1)) and int(ms.sum()) > 0
    for b in range(2):
        assert torch.equal(stu[b][ms[b]], tea[b][mt[b]])            # the answer tokens and the EOS, identical in both rows
    assert stu[0][ms[0]].tolist()[-1] == EOS and len(stu[0][ms[0]]) == 4   # "a small dog" + EOS


def test_schedule_helpers_match_transformers_and_reference_errors():
    """lr schedule == transformers.get_cosine_schedule_with_warmup (ref:icv_src/icv_module.py:189-209); type errors as the
    reference raises them (ref :66-67, :199-202).  Methods are exercised unbound on a stub (no engine needed)."""
    import types
    import torch
    from transformers import get_cosine_schedule_with_warmup
    from icv_src.icv_module import VQAICVModule
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sched = get_cosine_schedule_with_warmup(opt, num_warmup_steps=7, num_training_steps=50)
    for step in range(55):
        assert abs(sched.get_last_lr()[0] - VQAICVModule.lr_lambda(step, 7, 50)) <= 1e-12
        opt.step(); sched.step()
    cfg = types.SimpleNamespace(warm_steps=0.1, alpha_lr=1e-2, icv_lr=1e-4, weight_decay=1e-3, decay_per_step=-1, decay_ratio=-1)
    stub = types.SimpleNamespace(module_cfg=cfg)
    spec = VQAICVModule.optimizer_spec(stub, 200)
    assert spec == dict(alpha_lr=1e-2, icv_lr=1e-4, weight_decay=1e-3, warm_steps=20.0, total_steps=200)
    cfg.warm_steps = 15
    assert VQAICVModule.optimizer_spec(stub, 200)["warm_steps"] == 15
    cfg.warm_steps = "10"
    with pytest.raises(ValueError, match="warm_steps should be int or float"):
        VQAICVModule.optimizer_spec(stub, 200)
    cfg.decay_per_step = 1.5
    with pytest.raises(ValueError, match="decay_ratio must be an int or a float"):
        VQAICVModule.setup_temperature_decay(stub, 100)
    cfg.decay_per_step = 0.25
    VQAICVModule.setup_temperature_decay(stub, 100)
    assert stub.decay_per_step == 25


def test_temperature_decay_schedule():
    """decay_temperature (ref:icv_src/icv_module.py:150-158): T <- max(T * ratio, min) every decay_per_step optimiser steps."""
    import types
    import torch
    from icv_src.icv_module import VQAICVModule
    cfg = types.SimpleNamespace(decay_ratio=0.5, decay_per_step=2, min_tmeprature=0.3)
    stub = types.SimpleNamespace(module_cfg=cfg, temperature=torch.nn.Parameter(torch.tensor(2.0), requires_grad=False), global_step=0)
    VQAICVModule.setup_temperature_decay(stub, 100)
    seen = []
    for step in range(7):
        stub.global_step = step
        VQAICVModule.decay_temperature(stub)
        seen.append(round(float(stub.temperature), 4))
    assert seen == [2.0, 2.0, 1.0, 1.0, 0.5, 0.5, 0.3]
    cfg.decay_ratio = -1
    VQAICVModule.decay_temperature(stub)
    assert float(stub.temperature) == pytest.approx(0.3)


def test_encoder_output_is_the_reference_dataclass_shape():
    """ref:icv_src/icv_encoder/base_icv_encoder.py:7-11 — a mutable @dataclass with three positional fields."""
    import dataclasses
    assert dataclasses.is_dataclass(ICVEncoderOutput)
    assert [f.name for f in dataclasses.fields(ICVEncoderOutput)] == ["in_context_feature", "in_context_vector", "alpha"]
    out = ICVEncoderOutput(None, torch.zeros(1, 2, 4), torch.ones(1, 2))
    out.alpha = out.alpha * 2                                   # attribute assignment works (a NamedTuple would refuse)
    assert float(out.alpha.sum()) == 4.0
    with pytest.raises(TypeError):
        ICVEncoderOutput()                                      # no defaults, like the reference


def test_wrapper_exposes_the_reference_methods():
    w = LearnableICVInterventionLMM(torch.nn.Identity(), True, [1], "l.<LAYER_NUM>", 2)
    for name in ("apply_icv_intervention", "_get_context_manager", "_prepare_layers", "toggle_intervention", "forward", "generate"):
        assert callable(getattr(w, name)), name
    fn = w.apply_icv_intervention(w.intervention_layer_names, torch.zeros(1, 1, 4))
    x = torch.randn(2, 3, 4)
    assert fn(x, "l.0") is x                                    # a layer that is not edited passes through untouched (CPU ok)
