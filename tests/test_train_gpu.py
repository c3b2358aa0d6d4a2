"""Training step on the GPU: gradients of the reference's objective w.r.t. icv / alpha through the native backward
(HIP kernels) against the fixture produced by the reference's own VQAICVModule.forward + torch autograd (g6), then
the optimiser step against the oracle's AdamW."""
import numpy as np
import pytest
import torch

from licv.config import IDEFICS_TINY
from licv.synthetic import synth_idefics_weights
from oracle import icv_ref as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda"
FMT = "model.model.layers.<LAYER_NUM>"


def _module(temp, sd):
    from icv_src.icv_module import VQAICVModule
    from lmm_icl_interface import IdeficsInterface
    arch = IDEFICS_TINY
    iface = IdeficsInterface(state_dict=sd, arch=arch, device=DEV)
    mod_cfg = dict(hard_loss_weight=0.0, only_hard_loss=False, kl_eps=1e-6, init_temperature=temp, learnable_t=False,
                   decay_ratio=-1, decay_per_step=-1, min_tmeprature=1.0, alpha_lr=1e-2, icv_lr=1e-4, weight_decay=1e-3,
                   warm_steps=0.1, icv_encoder=dict(use_sigmoid=True, alpha_learnable=True, alpha_init_value=0.3))
    lmm_cfg = dict(intervention_layer=-1, layer_format=FMT, total_layers=arch.num_layers, hidden_size=arch.hidden_size)
    return VQAICVModule(iface, mod_cfg, lmm_cfg).to(DEV)


def _batch(z, prefix):
    return {k: T(z[f"{prefix}{k}"]) for k in ("input_ids", "attention_mask", "pixel_values", "image_attention_mask")}


@pytest.mark.parametrize("temp", [1.0, 2.0])
def test_native_backward_matches_reference_autograd(golden, temp):
    from licv.trainer import ICVTrainer
    z = golden("g6_loss")
    sd = synth_idefics_weights(IDEFICS_TINY, seed=31, dtype=torch.float32)
    mod = _module(temp, sd)
    with torch.no_grad():
        mod.icv_encoder.icv.copy_(T(z["enc_icv"]))
        mod.icv_encoder.alpha.copy_(T(z["enc_alpha_param"]))
    tr = ICVTrainer(mod, sd, total_steps=20, accumulate_grad_batches=1, grad_clip=1.0)
    kl = tr.loss_and_backward(_batch(z, "stu_"), _batch(z, "tea_"), T(z["query_x_length"]), T(z["in_context_length"]))
    key = f"T{int(temp)}"
    kl16, kl32 = float(z[f"bf16_{key}_kl"]), float(z[f"f32_{key}_kl"])
    assert abs(float(kl) - kl32) <= 1.5 * abs(kl16 - kl32) + 0.05 * kl32
    for name, got in (("grad_icv", mod.icv_encoder.icv.grad), ("grad_alpha", mod.icv_encoder.alpha.grad)):
        g32, g16 = T(z[f"f32_{key}_{name}"]), T(z[f"bf16_{key}_{name}"])
        spread = (g16 - g32).abs().max()
        err = (got.cpu() - g32).abs().max()
        scale = g32.abs().max()
        # held to the reference's own bf16-vs-fp32 gradient spread (plus 2 % of the gradient scale)
        assert err <= 1.5 * spread + 0.02 * scale, f"{name}: err {err:.3e} spread {spread:.3e} scale {scale:.3e}"
        cos = torch.nn.functional.cosine_similarity(got.cpu().reshape(1, -1), g32.reshape(1, -1)).item()
        assert cos > 0.99, f"{name}: cosine {cos}"


def test_trainer_step_applies_clipped_adamw_like_the_oracle(golden):
    from licv.trainer import ICVTrainer
    z = golden("g6_loss")
    sd = synth_idefics_weights(IDEFICS_TINY, seed=31, dtype=torch.float32)
    mod = _module(1.0, sd)
    with torch.no_grad():
        mod.icv_encoder.icv.copy_(T(z["enc_icv"]))
        mod.icv_encoder.alpha.copy_(T(z["enc_alpha_param"]))
    tr = ICVTrainer(mod, sd, total_steps=20, accumulate_grad_batches=2, grad_clip=1.0)
    p0 = torch.cat([mod.icv_encoder.alpha.detach().reshape(-1), mod.icv_encoder.icv.detach().reshape(-1)]).cpu()
    args = (_batch(z, "stu_"), _batch(z, "tea_"), T(z["query_x_length"]), T(z["in_context_length"]))
    assert tr.micro_batch(*args) is None                      # first micro-batch of the window: no optimiser step
    g_acc = torch.cat([mod.icv_encoder.alpha.grad.reshape(-1), mod.icv_encoder.icv.grad.reshape(-1)]).cpu().clone()
    log = tr.micro_batch(*args)                               # closes the window
    assert log is not None and set(log) >= {"kl_loss", "loss", "grad_norm"}
    # same micro-batch twice with upstream 1/2 each -> accumulated grad == 2 * first half
    g = 2 * g_acc
    (gc,), norm = O.clip_grad_norm([g], 1.0)
    assert abs(norm - log["grad_norm"]) <= 1e-3 * norm
    # step 0 of the cosine warm-up has lr 0 (LambdaLR): parameters unchanged, like torch's scheduler
    p1 = torch.cat([mod.icv_encoder.alpha.detach().reshape(-1), mod.icv_encoder.icv.detach().reshape(-1)]).cpu()
    assert torch.equal(p0, p1) and log["lr_scale"] == 0.0
    for _ in range(2):
        log = tr.micro_batch(*args)
    lam = O.cosine_warmup_lambda(1, 2.0, 20)
    assert abs(log["lr_scale"] - lam) < 1e-12 and lam == 0.5
    p2 = torch.cat([mod.icv_encoder.alpha.detach().reshape(-1), mod.icv_encoder.icv.detach().reshape(-1)]).cpu()
    assert (p2 - p1).abs().max() > 0
    # first real AdamW step: |delta| ~ lr * (1 / (1 + eps')) per coordinate with non-zero grad, sign opposite to the grad
    n_a = mod.icv_encoder.alpha.numel()
    moved = (p2 - p1)[n_a:]
    nz = g[n_a:].abs() > 1e-9
    assert (torch.sign(moved[nz]) == -torch.sign(g[n_a:][nz])).float().mean() > 0.99


@pytest.mark.parametrize("temp", [1.0, 2.0])
def test_idefics2_native_backward_matches_reference_autograd(golden, temp):
    """Idefics2 (hook on every `.mlp` branch, GQA attention backward): native grads vs the reference's VQAICVModule.forward +
    autograd through HF Idefics2 (fixture g9; bf16 = autocast).  Same bar as the Idefics test above."""
    from icv_src.icv_module import VQAICVModule
    from licv.config import IDEFICS2_TINY
    from licv.synthetic import synth_idefics2_weights
    from licv.trainer import ICVTrainer
    from lmm_icl_interface import Idefics2Interface
    z = golden("g9_loss_idefics2")
    arch = IDEFICS2_TINY
    sd = synth_idefics2_weights(arch, seed=91, dtype=torch.float32)
    iface = Idefics2Interface(state_dict=sd, arch=arch, device=DEV)
    mod_cfg = dict(hard_loss_weight=0.0, only_hard_loss=False, kl_eps=1e-6, init_temperature=temp, learnable_t=False,
                   decay_ratio=-1, decay_per_step=-1, min_tmeprature=1.0, alpha_lr=1e-2, icv_lr=1e-4, weight_decay=1e-3,
                   warm_steps=0.1, icv_encoder=dict(use_sigmoid=True, alpha_learnable=True, alpha_init_value=0.3))
    lmm_cfg = dict(intervention_layer=-1, layer_format="model.model.text_model.layers.<LAYER_NUM>.mlp", total_layers=arch.num_layers,
                   hidden_size=arch.hidden_size)
    mod = VQAICVModule(iface, mod_cfg, lmm_cfg).to(DEV)
    with torch.no_grad():
        mod.icv_encoder.icv.copy_(T(z["enc_icv"]))
        mod.icv_encoder.alpha.copy_(T(z["enc_alpha_param"]))
    tr = ICVTrainer(mod, sd, total_steps=20, accumulate_grad_batches=1, grad_clip=1.0)
    b = lambda p: {k: T(z[f"{p}{k}"]) for k in ("input_ids", "attention_mask", "pixel_values", "pixel_attention_mask")}
    kl = tr.loss_and_backward(b("stu_"), b("tea_"), T(z["query_x_length"]), T(z["in_context_length"]))
    key = f"T{int(temp)}"
    kl16, kl32 = float(z[f"bf16_{key}_kl"]), float(z[f"f32_{key}_kl"])
    assert abs(float(kl) - kl32) <= 1.5 * abs(kl16 - kl32) + 0.05 * kl32
    for name, got in (("grad_icv", mod.icv_encoder.icv.grad), ("grad_alpha", mod.icv_encoder.alpha.grad)):
        g32, g16 = T(z[f"f32_{key}_{name}"]), T(z[f"bf16_{key}_{name}"])
        spread = (g16 - g32).abs().max()
        err = (got.cpu() - g32).abs().max()
        scale = g32.abs().max()
        assert err <= 1.5 * spread + 0.02 * scale, f"{name}: err {err:.3e} spread {spread:.3e} scale {scale:.3e}"
        cos = torch.nn.functional.cosine_similarity(got.cpu().reshape(1, -1), g32.reshape(1, -1)).item()
        assert cos > 0.99, f"{name}: cosine {cos}"
    # the module's own forward (reference call shape) gives the same KL as the trainer's path
    loss_dict, _ = mod({k: v.to(DEV) for k, v in b("stu_").items()}, {k: v.to(DEV) for k, v in b("tea_").items()},
                       T(z["query_x_length"]).to(DEV), T(z["in_context_length"]).to(DEV))
    assert abs(float(loss_dict["kl_loss"]) - float(kl)) <= 2e-2 * abs(float(kl)) + 1e-5


def test_hard_loss_backward_matches_reference_autograd(golden):
    """loss = kl + 0.5 * ce (ref:icv_src/icv_module.py:94-95,111-117): CE rows forward/backward kernels + the KL path, against the
    reference module driving HF with labels (fixture g10: additional_vocab_size 0, full rows)."""
    from icv_src.icv_module import VQAICVModule
    from licv.trainer import ICVTrainer
    from lmm_icl_interface import IdeficsInterface
    z = golden("g10_hard_loss")
    arch = IDEFICS_TINY.with_(additional_vocab_size=0)
    sd = synth_idefics_weights(arch, seed=101, dtype=torch.float32)
    iface = IdeficsInterface(state_dict=sd, arch=arch, device=DEV)
    mod_cfg = dict(hard_loss_weight=float(z["hard_loss_weight"]), only_hard_loss=False, kl_eps=1e-6, init_temperature=1.0, learnable_t=False,
                   decay_ratio=-1, decay_per_step=-1, min_tmeprature=1.0, alpha_lr=1e-2, icv_lr=1e-4, weight_decay=1e-3,
                   warm_steps=0.1, icv_encoder=dict(use_sigmoid=True, alpha_learnable=True, alpha_init_value=0.3))
    lmm_cfg = dict(intervention_layer=-1, layer_format=FMT, total_layers=arch.num_layers, hidden_size=arch.hidden_size)
    mod = VQAICVModule(iface, mod_cfg, lmm_cfg).to(DEV)
    with torch.no_grad():
        mod.icv_encoder.icv.copy_(T(z["enc_icv"]))
        mod.icv_encoder.alpha.copy_(T(z["enc_alpha_param"]))
    tr = ICVTrainer(mod, sd, total_steps=20, accumulate_grad_batches=1, grad_clip=1.0)
    stu, tea = _batch(z, "stu_"), _batch(z, "tea_")
    kl = tr.loss_and_backward(stu, tea, T(z["query_x_length"]), T(z["in_context_length"]))
    assert abs(float(kl) - float(z["f32_kl"])) <= 1.5 * abs(float(z["bf16_kl"]) - float(z["f32_kl"])) + 0.05 * float(z["f32_kl"])
    assert abs(float(tr.last_ce) - float(z["f32_ce"])) <= 1.5 * abs(float(z["bf16_ce"]) - float(z["f32_ce"])) + 5e-3 * float(z["f32_ce"])
    for name, got in (("grad_icv", mod.icv_encoder.icv.grad), ("grad_alpha", mod.icv_encoder.alpha.grad)):
        g32, g16 = T(z[f"f32_{name}"]), T(z[f"bf16_{name}"])
        spread, err, scale = (g16 - g32).abs().max(), (got.cpu() - g32).abs().max(), g32.abs().max()
        assert err <= 1.5 * spread + 0.02 * scale, f"{name}: err {err:.3e} spread {spread:.3e} scale {scale:.3e}"
        assert torch.nn.functional.cosine_similarity(got.cpu().reshape(1, -1), g32.reshape(1, -1)).item() > 0.99
    # the module's forward (the reference's call shape) reports the same pieces
    loss_dict, _ = mod({k: v.to(DEV) for k, v in stu.items()}, {k: v.to(DEV) for k, v in tea.items()},
                       T(z["query_x_length"]).to(DEV), T(z["in_context_length"]).to(DEV))
    assert abs(float(loss_dict["ce_loss"]) - float(tr.last_ce)) <= 1e-3 * float(tr.last_ce)
    assert abs(float(loss_dict["loss"]) - (float(loss_dict["kl_loss"]) + 0.5 * float(loss_dict["ce_loss"]))) <= 1e-5


def test_only_hard_loss_gradient_matches_oracle_autograd(golden):
    """only_hard_loss (ref:icv_src/icv_module.py:100-101): the objective is the student's CE alone; checked against torch autograd
    through the CPU oracle on the g10 inputs."""
    from icv_src.icv_module import VQAICVModule
    from licv.trainer import ICVTrainer
    from lmm_icl_interface import IdeficsInterface
    from oracle import idefics_ref as R
    z = golden("g10_hard_loss")
    arch = IDEFICS_TINY.with_(additional_vocab_size=0)
    sd = synth_idefics_weights(arch, seed=101, dtype=torch.float32)
    iface = IdeficsInterface(state_dict=sd, arch=arch, device=DEV)
    mod_cfg = dict(hard_loss_weight=1.0, only_hard_loss=True, kl_eps=1e-6, init_temperature=1.0, learnable_t=False,
                   decay_ratio=-1, decay_per_step=-1, min_tmeprature=1.0, alpha_lr=1e-2, icv_lr=1e-4, weight_decay=1e-3,
                   warm_steps=0.1, icv_encoder=dict(use_sigmoid=True, alpha_learnable=True, alpha_init_value=0.3))
    lmm_cfg = dict(intervention_layer=-1, layer_format=FMT, total_layers=arch.num_layers, hidden_size=arch.hidden_size)
    mod = VQAICVModule(iface, mod_cfg, lmm_cfg).to(DEV)
    with torch.no_grad():
        mod.icv_encoder.icv.copy_(T(z["enc_icv"]))
        mod.icv_encoder.alpha.copy_(T(z["enc_alpha_param"]))
    tr = ICVTrainer(mod, sd, total_steps=20, accumulate_grad_batches=1, grad_clip=1.0)
    stu = _batch(z, "stu_")
    ce = tr.loss_and_backward(stu, _batch(z, "tea_"), T(z["query_x_length"]), T(z["in_context_length"]))
    icv = T(z["enc_icv"]).clone().requires_grad_(True)
    alpha = T(z["enc_alpha_param"]).clone().requires_grad_(True)
    lg = R.forward(sd, arch, **stu, icv=O.scale_icv(O.encoder_alpha(alpha, True), icv), hook_layers=list(range(arch.num_layers)))
    ref = O.ce_masked(lg, stu["input_ids"], stu["attention_mask"])
    ref.backward()
    assert abs(float(ce) - float(ref)) <= 5e-3 * float(ref)
    for got, want in ((mod.icv_encoder.icv.grad.cpu(), icv.grad), (mod.icv_encoder.alpha.grad.cpu(), alpha.grad)):
        assert torch.nn.functional.cosine_similarity(got.reshape(1, -1), want.reshape(1, -1)).item() > 0.99
        assert (got - want).abs().max() <= 0.05 * want.abs().max()
