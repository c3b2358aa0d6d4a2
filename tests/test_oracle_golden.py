"""The CPU oracle against the fixtures produced by the reference itself (tools/make_golden.py)."""
import numpy as np
import pytest
import torch

from licv.config import IDEFICS2_MID, IDEFICS2_TINY, IDEFICS_MID, IDEFICS_TINY
from licv.synthetic import synth_idefics2_weights, synth_idefics_weights, weights_checksum
from oracle import icv_ref as O
from oracle import idefics_ref as R
from oracle import idefics2_ref as R2

T = torch.from_numpy


def test_g1_encoder_init_and_alpha(golden):
    z = golden("g1_encoder")
    for tag in ("a", "b"):
        H, L, a0, sig = z[f"{tag}_cfg"]
        torch.manual_seed(426)
        icv, alpha = O.encoder_init(int(H), int(L), float(a0))
        assert torch.equal(icv, T(z[f"{tag}_icv"]))
        assert torch.equal(alpha, T(z[f"{tag}_alpha_param"]))
        assert torch.equal(O.encoder_alpha(alpha, bool(sig)), T(z[f"{tag}_alpha_out"]))
        assert list(z[f"{tag}_state_keys"]) == ["alpha", "icv"]


def test_g2_layer_bookkeeping(golden):
    z = golden("g2_intervention")
    layers = O.prepare_layers([3, 7, 1], 8)
    assert O.layer_names(layers, "model.model.layers.<LAYER_NUM>") == list(z["names"])
    m = O.layer_to_icv_index(layers)
    assert list(m.keys()) == list(z["map_keys"]) and list(m.values()) == list(z["map_vals"])
    assert O.layer_names(O.prepare_layers(-1, 5), "blk.<LAYER_NUM>.mlp") == list(z["names_all"])
    assert O.layer_names(O.prepare_layers(2, 5), "blk.<LAYER_NUM>") == list(z["names_int"])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_g2_inject_renorm(golden, dt):
    z = golden("g2_intervention")
    icv = T(z["icv"])
    h = T(z[f"h_{dt}"]).to(torch.bfloat16 if dt == "bf16" else torch.float32)
    m = O.layer_to_icv_index([3, 7, 1])
    out = O.inject_renorm(h, icv[:, m[7]])
    assert out.dtype == torch.float32 and bool(z[f"tensor_{dt}_is_f32"])
    assert torch.equal(out, T(z[f"tensor_{dt}"]))
    assert torch.equal(O.inject_renorm(h, icv[:, m[1]]), T(z[f"tuple_{dt}"]))


def test_inject_renorm_bwd_matches_autograd():
    g = torch.Generator().manual_seed(0)
    h = torch.randn(3, 5, 32, generator=g, dtype=torch.float64, requires_grad=True)
    v = torch.randn(32, generator=g, dtype=torch.float64, requires_grad=True)
    go = torch.randn(3, 5, 32, generator=g, dtype=torch.float64)
    O.inject_renorm(h, v).backward(go)
    gh, gv = O.inject_renorm_bwd(h.detach(), v.detach(), go)
    assert torch.allclose(gh, h.grad, atol=1e-12) and torch.allclose(gv, v.grad, atol=1e-12)


@pytest.mark.parametrize("tag,arch", [("g3_idefics_tiny", IDEFICS_TINY), ("g3_idefics_mid", IDEFICS_MID)])
@pytest.mark.parametrize("dn,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_g3_idefics_forward(golden, tag, arch, dn, dt):
    z = golden(tag)
    sd32 = synth_idefics_weights(arch, seed=int(z["meta"][0]), dtype=torch.float32)
    assert weights_checksum(sd32) == float(z["weights_checksum"]), "seeded weight generator drifted"
    sd = {k: v.to(dt) for k, v in sd32.items()}
    ins = dict(input_ids=T(z["in_input_ids"]), attention_mask=T(z["in_attention_mask"]),
               pixel_values=T(z["in_pixel_values"]).to(dt), image_attention_mask=T(z["in_image_attention_mask"]))
    tol = 1e-5 if dn == "f32" else 0.0          # bf16 restatement is bit-for-bit the HF path on CPU
    with torch.no_grad():
        cap = {}
        off = R.forward(sd, arch, **ins, capture=cap)
        assert (off.float() - T(z[f"{dn}_logits_off"])).abs().max() <= tol
        assert (cap["image_states"].float().reshape(-1) - T(z[f"{dn}_image_states"]).reshape(-1)).abs().max() <= tol
        for hs, layers in (("all", list(range(arch.num_layers))), ("sub", [1, 3])):
            if f"{dn}_{hs}_logits" not in z.files:
                continue
            icv = T(z["icv_full"])[:, : len(layers)]
            cap = {}
            lg = R.forward(sd, arch, **ins, icv=icv, hook_layers=layers, capture=cap)
            assert (lg.float() - T(z[f"{dn}_{hs}_logits"])).abs().max() <= tol
            raw = torch.stack([t.float() for t in cap["raw"]])
            assert (raw - T(z[f"{dn}_{hs}_raw"])).abs().max() <= tol
            ed = torch.stack([cap["edited"][i].float() for i in z[f"{dn}_{hs}_edited_idx"]])
            assert (ed - T(z[f"{dn}_{hs}_edited"])).abs().max() <= tol
            # fp32 icv promotes the residual stream to fp32 from the first hooked layer on
            assert cap["edited"][-1].dtype == torch.float32


@pytest.mark.parametrize("tag,arch", [("g4_idefics2_tiny", IDEFICS2_TINY), ("g4_idefics2_mid", IDEFICS2_MID)])
@pytest.mark.parametrize("dn,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_g4_idefics2_forward(golden, tag, arch, dn, dt):
    """Idefics2 (hook on every text layer's MLP branch) against HF driven by the reference's wrapper.  bf16 = the
    reference's autocast regime; the oracle runs inside the same autocast context."""
    import contextlib
    z = golden(tag)
    sd32 = synth_idefics2_weights(arch, seed=int(z["meta"][0]), dtype=torch.float32)
    assert weights_checksum(sd32) == float(z["weights_checksum"]), "seeded weight generator drifted"
    sd = {k: v.to(dt) for k, v in sd32.items()}
    ins = dict(input_ids=T(z["in_input_ids"]), attention_mask=T(z["in_attention_mask"]),
               pixel_values=T(z["in_pixel_values"]).to(dt), pixel_attention_mask=T(z["in_pixel_attention_mask"]))
    tol = 1e-5 if dn == "f32" else 0.0
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if dn == "bf16" else contextlib.nullcontext()
    layers = list(range(arch.num_layers))
    with torch.no_grad(), ctx:
        cap = {}
        off = R2.forward(sd, arch, **ins, capture=cap)
        assert (off.float() - T(z[f"{dn}_logits_off"])).abs().max() <= tol
        assert (cap["image_hidden_states"].float() - T(z[f"{dn}_image_hidden_states"]).reshape(cap["image_hidden_states"].shape)).abs().max() <= tol
        cap = {}
        lg = R2.forward(sd, arch, **ins, icv=T(z["icv_full"]), hook_layers=layers, capture=cap)
        assert (lg.float() - T(z[f"{dn}_all_logits"])).abs().max() <= tol
        assert (torch.stack([t.float() for t in cap["mlp_raw"]]) - T(z[f"{dn}_all_mlp_raw"])).abs().max() <= tol
        assert (torch.stack([t.float() for t in cap["layer_out"]]) - T(z[f"{dn}_all_layer_out"])).abs().max() <= tol
        assert cap["layer_out"][0].dtype == torch.float32          # the hooked branch promotes the stream to fp32
        assert bool(z[f"{dn}_layer_out_is_f32"])


def _g6_setup(golden, dt):
    z = golden("g6_loss")
    arch = IDEFICS_TINY
    sd32 = synth_idefics_weights(arch, seed=31, dtype=torch.float32)
    assert weights_checksum(sd32) == float(z["weights_checksum"])
    sd = {k: v.to(dt) for k, v in sd32.items()}
    def b(n):
        return dict(input_ids=T(z[f"{n}_input_ids"]), attention_mask=T(z[f"{n}_attention_mask"]),
                    pixel_values=T(z[f"{n}_pixel_values"]).to(dt), image_attention_mask=T(z[f"{n}_image_attention_mask"]))
    return z, arch, sd, b("stu"), b("tea")


def test_g6_masks(golden):
    z, arch, sd, stu, tea = _g6_setup(golden, torch.float32)
    assert torch.equal(O.get_mask(stu["input_ids"], T(z["query_x_length"]), arch.pad_token_id), T(z["stu_mask"]))
    assert torch.equal(O.get_mask(tea["input_ids"], T(z["in_context_length"]), arch.pad_token_id), T(z["tea_mask"]))
    assert int(T(z["stu_mask"]).sum()) == int(T(z["tea_mask"]).sum())


@pytest.mark.parametrize("dn,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("temp", [1.0, 2.0])
def test_g6_kl_loss_and_grads(golden, dn, dt, temp):
    z, arch, sd, stu, tea = _g6_setup(golden, dt)
    icv = T(z["enc_icv"]).clone().requires_grad_(True)
    alpha = T(z["enc_alpha_param"]).clone().requires_grad_(True)
    icv_eff = O.scale_icv(O.encoder_alpha(alpha, True), icv)
    layers = list(range(arch.num_layers))
    s_logits = R.forward(sd, arch, **stu, icv=icv_eff, hook_layers=layers)
    with torch.no_grad():
        t_logits = R.forward(sd, arch, **tea)
    sm, tm = T(z["stu_mask"]), T(z["tea_mask"])
    kl = O.kl_divergence(s_logits[sm].view(-1, s_logits.shape[-1]), t_logits[tm].view(-1, t_logits.shape[-1]), temp)
    kl.backward()
    key = f"{dn}_T{int(temp)}"
    tol = 1e-6 if dn == "f32" else 0.0
    assert abs(float(kl) - float(z[f"{key}_kl"])) <= tol
    gtol = 1e-7 if dn == "f32" else 0.0
    assert (icv.grad - T(z[f"{key}_grad_icv"])).abs().max() <= gtol
    assert (alpha.grad - T(z[f"{key}_grad_alpha"])).abs().max() <= gtol


def test_g6_student_logits_and_ce(golden):
    z, arch, sd, stu, tea = _g6_setup(golden, torch.float32)
    icv_eff = O.scale_icv(O.encoder_alpha(T(z["enc_alpha_param"]), True), T(z["enc_icv"]))
    with torch.no_grad():
        lg = R.forward(sd, arch, **stu, icv=icv_eff, hook_layers=list(range(arch.num_layers)))
    assert (lg - T(z["f32_student_logits"])).abs().max() <= 1e-6
    ce = O.ce_masked(lg, stu["input_ids"], stu["attention_mask"])
    assert abs(float(ce) - float(z["f32_student_ce"])) <= 1e-5


def test_g7_adamw_cosine(golden):
    z = golden("g7_optim")
    icv, alpha = T(z["icv0"]).clone(), T(z["alpha0"]).clone()
    mi, vi, ma, va = [torch.zeros_like(x) for x in (icv, icv, alpha, alpha)]
    total, warm = float(z["total"]), float(z["warm"])
    for s in range(z["lrs"].shape[0]):
        lam = O.cosine_warmup_lambda(s, warm, total)
        lr_a, lr_i = 1e-2 * lam, 1e-4 * lam
        assert np.allclose([lr_a, lr_i], z["lrs"][s], rtol=1e-12, atol=0)
        (ga, gi), _ = O.clip_grad_norm([T(z["grads_alpha"][s]), T(z["grads_icv"][s])], 1.0)
        alpha, ma, va = O.adamw_step(alpha, ga, ma, va, s + 1, lr_a)
        icv, mi, vi = O.adamw_step(icv, gi, mi, vi, s + 1, lr_i)
        assert (icv - T(z["icv_steps"][s])).abs().max() <= 1e-8
        assert (alpha - T(z["alpha_steps"][s])).abs().max() <= 1e-7


@pytest.mark.parametrize("side", ["left", "right"])
def test_g5_generate(golden, side):
    """Oracle decode == ids the reference wrapper + HF generate produced (fp32): beam (3 beams, 5 tokens,
    length_penalty 0), greedy, and greedy with the intervention toggled off."""
    from oracle.generate_ref import generate
    z = golden("g5_generate")
    arch = IDEFICS_TINY.with_(additional_vocab_size=0)
    sd = synth_idefics_weights(arch, seed=21, dtype=torch.float32)
    assert weights_checksum(sd) == float(z["weights_checksum"])
    sd["model.embed_tokens.weight"] *= float(z["embed_scale"])
    sd["lm_head.weight"] *= float(z["head_scale"])
    b = {k: T(z[f"{side}_in_{k}"]) for k in ("input_ids", "attention_mask", "pixel_values", "image_attention_mask")}
    icv, layers = T(z["icv"]), list(range(arch.num_layers))
    kw = dict(max_new_tokens=5, length_penalty=0.0)
    assert torch.equal(generate(sd, arch, **b, icv=icv, hook_layers=layers, num_beams=3, **kw), T(z[f"{side}_f32_beam_ids"]))
    assert torch.equal(generate(sd, arch, **b, icv=icv, hook_layers=layers, num_beams=1, **kw), T(z[f"{side}_f32_greedy_ids"]))
    assert torch.equal(generate(sd, arch, **b, num_beams=1, **kw), T(z[f"{side}_f32_greedy_off_ids"]))


def _g8_setup(golden):
    z = golden("g8_generate_idefics2")
    arch = IDEFICS2_TINY
    sd = synth_idefics2_weights(arch, seed=81, dtype=torch.float32)
    assert weights_checksum(sd) == float(z["weights_checksum"])
    sd["model.text_model.embed_tokens.weight"] *= float(z["embed_scale"])
    sd["lm_head.weight"] *= float(z["head_scale"])
    for l in range(arch.num_layers):
        sd[f"model.text_model.layers.{l}.mlp.down_proj.weight"] *= float(z["down_scale"])
    return z, arch, sd


@pytest.mark.parametrize("side", ["left", "right"])
def test_g8_generate_idefics2(golden, side):
    """Oracle decode == ids the reference wrapper + HF Idefics2 generate produced (fp32, hook on every `.mlp`)."""
    from oracle.generate_ref import generate_idefics2
    z, arch, sd = _g8_setup(golden)
    b = {k: T(z[f"{side}_in_{k}"]) for k in ("input_ids", "attention_mask", "pixel_values", "pixel_attention_mask")}
    icv, layers = T(z["icv"]), list(range(arch.num_layers))
    kw = dict(max_new_tokens=5, length_penalty=0.0)
    beam = generate_idefics2(sd, arch, **b, icv=icv, hook_layers=layers, num_beams=3, **kw)
    assert torch.equal(beam, T(z[f"{side}_f32_beam_ids"]))
    greedy = generate_idefics2(sd, arch, **b, icv=icv, hook_layers=layers, num_beams=1, **kw)
    assert torch.equal(greedy, T(z[f"{side}_f32_greedy_ids"]))
    assert torch.equal(generate_idefics2(sd, arch, **b, num_beams=1, **kw), T(z[f"{side}_f32_greedy_off_ids"]))
    P = b["input_ids"].shape[1]
    assert not torch.equal(greedy[:, P:], T(z[f"{side}_f32_greedy_off_ids"])[:, P:]), "fixture does not exercise the hook"


def _g9_setup(golden, dt):
    z = golden("g9_loss_idefics2")
    arch = IDEFICS2_TINY
    sd32 = synth_idefics2_weights(arch, seed=91, dtype=torch.float32)
    assert weights_checksum(sd32) == float(z["weights_checksum"])
    sd = {k: v.to(dt) for k, v in sd32.items()}

    def b(n):
        return dict(input_ids=T(z[f"{n}_input_ids"]), attention_mask=T(z[f"{n}_attention_mask"]),
                    pixel_values=T(z[f"{n}_pixel_values"]).to(dt), pixel_attention_mask=T(z[f"{n}_pixel_attention_mask"]))
    return z, arch, sd, b("stu"), b("tea")


@pytest.mark.parametrize("dn,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("temp", [1.0, 2.0])
def test_g9_idefics2_kl_loss_and_grads(golden, dn, dt, temp):
    """Training objective on Idefics2 (hook on every `.mlp`): KL and d/d icv, d/d alpha against the reference's
    VQAICVModule.forward + autograd through HF (bf16 = autocast regime)."""
    import contextlib
    z, arch, sd, stu, tea = _g9_setup(golden, dt)
    icv = T(z["enc_icv"]).clone().requires_grad_(True)
    alpha = T(z["enc_alpha_param"]).clone().requires_grad_(True)
    layers = list(range(arch.num_layers))
    sm, tm = T(z["stu_mask"]), T(z["tea_mask"])
    assert torch.equal(O.get_mask(stu["input_ids"], T(z["query_x_length"]), arch.pad_token_id), sm)
    assert torch.equal(O.get_mask(tea["input_ids"], T(z["in_context_length"]), arch.pad_token_id), tm)
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if dn == "bf16" else contextlib.nullcontext()
    with ctx:
        icv_eff = O.scale_icv(O.encoder_alpha(alpha, True), icv)
        s_logits = R2.forward(sd, arch, **stu, icv=icv_eff, hook_layers=layers)
        with torch.no_grad():
            t_logits = R2.forward(sd, arch, **tea)
        kl = O.kl_divergence(s_logits[sm].view(-1, s_logits.shape[-1]), t_logits[tm].view(-1, t_logits.shape[-1]), temp)
    kl.backward()
    key = f"{dn}_T{int(temp)}"
    tol = 1e-6 if dn == "f32" else 0.0
    assert abs(float(kl) - float(z[f"{key}_kl"])) <= tol
    gtol = 1e-7 if dn == "f32" else 0.0
    assert (icv.grad - T(z[f"{key}_grad_icv"])).abs().max() <= gtol
    assert (alpha.grad - T(z[f"{key}_grad_alpha"])).abs().max() <= gtol


@pytest.mark.parametrize("dn,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_g10_hard_loss_and_grads(golden, dn, dt):
    """loss = kl + hard_loss_weight * ce with the reference's own forward (HF computes the CE): value and grads of icv / alpha."""
    z = golden("g10_hard_loss")
    arch = IDEFICS_TINY.with_(additional_vocab_size=0)
    sd32 = synth_idefics_weights(arch, seed=101, dtype=torch.float32)
    assert weights_checksum(sd32) == float(z["weights_checksum"])
    sd = {k: v.to(dt) for k, v in sd32.items()}
    b = lambda n: dict(input_ids=T(z[f"{n}_input_ids"]), attention_mask=T(z[f"{n}_attention_mask"]),
                       pixel_values=T(z[f"{n}_pixel_values"]).to(dt), image_attention_mask=T(z[f"{n}_image_attention_mask"]))
    stu, tea = b("stu"), b("tea")
    icv = T(z["enc_icv"]).clone().requires_grad_(True)
    alpha = T(z["enc_alpha_param"]).clone().requires_grad_(True)
    icv_eff = O.scale_icv(O.encoder_alpha(alpha, True), icv)
    layers = list(range(arch.num_layers))
    s_logits = R.forward(sd, arch, **stu, icv=icv_eff.detach() if False else icv_eff, hook_layers=layers)
    with torch.no_grad():
        t_logits = R.forward(sd, arch, **tea)
    sm = O.get_mask(stu["input_ids"], T(z["query_x_length"]), arch.pad_token_id)
    tm = O.get_mask(tea["input_ids"], T(z["in_context_length"]), arch.pad_token_id)
    kl = O.kl_divergence(s_logits[sm].view(-1, s_logits.shape[-1]), t_logits[tm].view(-1, t_logits.shape[-1]), 1.0)
    ce = O.ce_masked(s_logits, stu["input_ids"], stu["attention_mask"])
    loss = kl + float(z["hard_loss_weight"]) * ce
    loss.backward()
    tol = 2e-6 if dn == "f32" else 0.0
    assert abs(float(kl) - float(z[f"{dn}_kl"])) <= tol
    # the CE is an fp32 reduction over (rows, vocab) in both paths; HF sums then divides, the oracle takes the mean: 1 fp32 ulp
    assert abs(float(ce) - float(z[f"{dn}_ce"])) <= 2e-6 and abs(float(loss) - float(z[f"{dn}_loss"])) <= 2e-6
    g_icv, g_alpha = T(z[f"{dn}_grad_icv"]), T(z[f"{dn}_grad_alpha"])
    assert (icv.grad - g_icv).abs().max() <= (1e-7 if dn == "f32" else 2e-3 * g_icv.abs().max())
    assert (alpha.grad - g_alpha).abs().max() <= (1e-7 if dn == "f32" else 2e-3 * g_alpha.abs().max())


@pytest.mark.parametrize("model,stable", [("idefics", False), ("idefics2", False), ("idefics", True), ("idefics2", True)])
def test_oracle_bf16_decode_equals_reference_bf16_generate(golden, model, stable):
    """g11 / g12: the reference wrapper driving HF generate in its bf16 regime, 16 prompts per padding side, beams 3 / greedy /
    hooks off.  The oracle's decode (oracle/generate_ref.py) must reproduce every id: this is what pins the bf16 search path
    the GPU tests compare with.  g15 / g16 (`stable`): the same on the well-conditioned prompts (every row reproduced by all 72
    jittered re-runs of the reference) that the GPU test requires bit for bit."""
    import torch
    from oracle import generate_ref as G
    T = torch.from_numpy
    if model == "idefics":
        from licv.config import IDEFICS_TINY
        from licv.synthetic import synth_idefics_weights
        z = golden("g15_generate_bf16_stable" if stable else "g11_generate_bf16")
        arch = IDEFICS_TINY.with_(additional_vocab_size=0)
        sd = synth_idefics_weights(arch, seed=int(z["weights_seed"]) if stable else 121, dtype=torch.float32)
        sd["model.embed_tokens.weight"] *= float(z["embed_scale"])
        keys, gen, mask_key = ("input_ids", "attention_mask", "pixel_values", "image_attention_mask"), G.generate, None
    else:
        from licv.config import IDEFICS2_TINY
        from licv.synthetic import synth_idefics2_weights
        z = golden("g16_generate_idefics2_bf16_stable" if stable else "g12_generate_idefics2_bf16")
        arch = IDEFICS2_TINY
        sd = synth_idefics2_weights(arch, seed=int(z["weights_seed"]) if stable else 181, dtype=torch.float32)
        sd["model.text_model.embed_tokens.weight"] *= float(z["embed_scale"])
        for l in range(arch.num_layers):
            sd[f"model.text_model.layers.{l}.mlp.down_proj.weight"] *= float(z["down_scale"])
        keys, gen = ("input_ids", "attention_mask", "pixel_values", "pixel_attention_mask"), G.generate_idefics2
    sd["lm_head.weight"] *= float(z["head_scale"])
    sd = {k: v.to(torch.bfloat16) for k, v in sd.items()}
    icv, layers = T(z["icv"]), list(range(arch.num_layers))
    kw = dict(max_new_tokens=5, length_penalty=0.0, min_new_tokens=0)
    import contextlib
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if model == "idefics2" else contextlib.nullcontext()
    for side in ("left", "right"):
        cb = {k: T(z[f"{side}_in_{k}"]) for k in keys}
        cb["pixel_values"] = cb["pixel_values"].to(torch.bfloat16)
        with ctx:
            outs = dict(beam=gen(sd, arch, **cb, icv=icv, hook_layers=layers, num_beams=3, **kw),
                        greedy=gen(sd, arch, **cb, icv=icv, hook_layers=layers, num_beams=1, **kw),
                        greedy_off=gen(sd, arch, **cb, num_beams=1, **kw))
        for tag, got in outs.items():
            assert torch.equal(got, T(z[f"{side}_bf16_{tag}_ids"])), f"{model} {side} {tag}"
            if stable:
                assert bool((T(z[f"{side}_bf16_{tag}_stability"]) >= 1.0).all())


def test_oracle_frontend_rules_match_hf_fixture(golden):
    """g13: transformers' own image_attention_mask functions and the patch mask / NaViT position ids captured inside HF Idefics2."""
    import torch
    from oracle import frontend_ref as F
    from oracle import idefics2_ref as R2
    T = torch.from_numpy
    z = golden("g13_frontend")
    for tag in "abcd":
        img, eod, n = (int(v) for v in z[f"m_{tag}_cfg"])
        got = F.image_attention_mask(T(z[f"m_{tag}_ids"]), img, eod, n)
        assert torch.equal(got, T(z[f"m_{tag}_mask"]).long()), tag
    for tag, patch, n_side in (("tiny", 14, 4), ("mid", 14, 6)):
        pv, pam = T(z[f"v_{tag}_pixel_values"]), T(z[f"v_{tag}_pixel_attention_mask"])
        B, N = pv.shape[:2]
        pvf = pv.reshape(B * N, *pv.shape[2:]).to(torch.bfloat16)
        real = (pvf == 0.0).sum(dim=(-1, -2, -3)) != pvf.shape[1:].numel()
        assert int(real.sum()) == int(z[f"v_{tag}_n_real"])
        pm = R2.patch_mask_from_pixels(pam.reshape(B * N, *pam.shape[2:])[real], patch)
        assert torch.equal(pm, T(z[f"v_{tag}_patch_mask"]))
        fh, fw, bounds = R2.navit_position_ids(pm, n_side)
        bh = torch.bucketize(fh.to(torch.bfloat16), bounds, right=True)
        bw = torch.bucketize(fw.to(torch.bfloat16), bounds, right=True)
        pos = (bh[:, :, None] * n_side + bw[:, None, :]).reshape(pm.shape[0], -1)
        pos = torch.where(pm.view(pm.shape[0], -1), pos, torch.zeros_like(pos))
        assert torch.equal(pos, T(z[f"v_{tag}_position_ids"]))


def test_oracle_image_preprocess_matches_hf_image_processor(golden):
    """g17: HF's IdeficsImageProcessorPil on seeded uint8 images (resize off); transformers.image_transforms rescale / normalize with
    the Idefics2 mean / std and padding rule.  The oracle's numpy restatement reproduces both bit for bit in float32."""
    import numpy as np
    from oracle import frontend_ref as F
    z = golden("g17_image_preprocess")
    y, m = F.preprocess_images(z["idefics_u8"], z["idefics_mean"], z["idefics_std"], float(z["idefics_rescale"]))
    assert np.array_equal(y, z["idefics_f32"]) and int(m.min()) == 1
    u = z["idefics2_u8"]
    B, N, H, W = u.shape[:4]
    y2, m2 = F.preprocess_images(u.reshape(B * N, H, W, 3), (0.5,) * 3, (0.5,) * 3, 1 / 255, z["idefics2_hw"].reshape(B * N, 2))
    assert np.array_equal(y2.reshape(B, N, 3, H, W), z["idefics2_f32"]) and np.array_equal(m2.reshape(B, N, H, W), z["idefics2_mask"])
