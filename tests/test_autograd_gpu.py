"""The reference's training call shape is differentiable on the native path (ref:icv_src/icv_module.py:97-118,160-209):
``VQAICVModule.forward(...)[0]["loss"].backward()`` fills ``icv.grad`` / ``alpha.grad`` through the explicit HIP backward
(licv.autograd), and ``training_step`` + ``configure_optimizers`` drive a step the Lightning way.  Checked against
ICVTrainer's explicit path (itself pinned to the reference's autograd by fixture g6, tests/test_train_gpu.py) and against
the reference's own gradients (g6, g10)."""
import pytest
import torch

from licv.config import IDEFICS_TINY
from licv.synthetic import synth_idefics_weights
from oracle import icv_ref as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda"
FMT = "model.model.layers.<LAYER_NUM>"


def _module(temp=1.0, hard_w=0.0, use_sigmoid=True, alpha0=0.3, arch=IDEFICS_TINY, seed=31):
    from icv_src.icv_module import VQAICVModule
    from lmm_icl_interface import IdeficsInterface
    sd = synth_idefics_weights(arch, seed=seed, dtype=torch.float32)
    iface = IdeficsInterface(state_dict=sd, arch=arch, device=DEV)
    mod_cfg = dict(hard_loss_weight=hard_w, only_hard_loss=False, kl_eps=1e-6, init_temperature=temp, learnable_t=False,
                   decay_ratio=-1, decay_per_step=-1, min_tmeprature=1.0, alpha_lr=1e-2, icv_lr=1e-4, weight_decay=1e-3,
                   warm_steps=0.1, log_alpha=True, strategy="ddp",
                   icv_encoder=dict(use_sigmoid=use_sigmoid, alpha_learnable=True, alpha_init_value=alpha0))
    lmm_cfg = dict(intervention_layer=-1, layer_format=FMT, total_layers=arch.num_layers, hidden_size=arch.hidden_size)
    return VQAICVModule(iface, mod_cfg, lmm_cfg).to(DEV)


def _load(mod, z):
    with torch.no_grad():
        mod.icv_encoder.icv.copy_(T(z["enc_icv"]))
        mod.icv_encoder.alpha.copy_(T(z["enc_alpha_param"]))


def _batch(z, prefix):
    return {k: T(z[f"{prefix}{k}"]).to(DEV) for k in ("input_ids", "attention_mask", "pixel_values", "image_attention_mask")}


def _args(z):
    return _batch(z, "stu_"), _batch(z, "tea_"), T(z["query_x_length"]).to(DEV), T(z["in_context_length"]).to(DEV)


@pytest.mark.parametrize("temp", [1.0, 2.0])
def test_loss_backward_equals_trainer_backward_bitwise_and_reference_autograd(golden, temp):
    from licv.trainer import ICVTrainer
    z = golden("g6_loss")
    a, b = _module(temp), _module(temp)
    _load(a, z); _load(b, z)
    loss_dict, _ = a(*_args(z))
    assert loss_dict["loss"].requires_grad, "the drop-in forward must return a differentiable loss"
    loss_dict["loss"].backward()
    ICVTrainer(b, total_steps=20, accumulate_grad_batches=1, grad_clip=1.0).loss_and_backward(*_args(z))
    for name in ("icv", "alpha"):
        ga, gb = getattr(a.icv_encoder, name).grad, getattr(b.icv_encoder, name).grad
        assert ga is not None and torch.equal(ga, gb), f"{name}: autograd path and explicit path must agree bit for bit"
    key = f"T{int(temp)}"
    for name, got in (("grad_icv", a.icv_encoder.icv.grad), ("grad_alpha", a.icv_encoder.alpha.grad)):
        g32, g16 = T(z[f"f32_{key}_{name}"]), T(z[f"bf16_{key}_{name}"])
        assert (got.cpu() - g32).abs().max() <= 1.5 * (g16 - g32).abs().max() + 0.02 * g32.abs().max()


def test_hard_loss_term_is_differentiable_and_close_to_explicit_path(golden):
    from licv.trainer import ICVTrainer
    z = golden("g10_hard_loss")
    w = float(z["hard_loss_weight"])
    arch = IDEFICS_TINY.with_(additional_vocab_size=0)                 # g10's model (HF's CE needs additional_vocab_size 0)
    a, b = _module(1.0, hard_w=w, arch=arch, seed=101), _module(1.0, hard_w=w, arch=arch, seed=101)
    _load(a, z); _load(b, z)
    loss_dict, _ = a(*_args(z))
    assert set(loss_dict) == {"kl_loss", "ce_loss", "loss"}
    loss_dict["loss"].backward()
    ICVTrainer(b, total_steps=20, accumulate_grad_batches=1, grad_clip=1.0).loss_and_backward(*_args(z))
    for name in ("icv", "alpha"):
        ga, gb = getattr(a.icv_encoder, name).grad, getattr(b.icv_encoder, name).grad
        # the two paths round the summed logit gradient (KL + w*CE) to bf16 at different points: close, not bitwise
        assert (ga - gb).abs().max() <= 2e-2 * gb.abs().max() + 1e-9
        assert torch.nn.functional.cosine_similarity(ga.reshape(1, -1), gb.reshape(1, -1)).item() > 0.999


def test_training_step_and_configure_optimizers_match_trainer_steps(golden):
    """Three Lightning-style steps (training_step -> backward -> optimizer.step -> scheduler.step) == three ICVTrainer
    steps without clipping: same gradients, same fused AdamW arithmetic, same cosine warm-up values."""
    from licv.trainer import ICVTrainer
    z = golden("g6_loss")
    a, b = _module(), _module()
    _load(a, z); _load(b, z)
    cfg = a.configure_optimizers(estimated_stepping_batches=20)
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    assert cfg["lr_scheduler"]["interval"] == "step" and len(opt.param_groups) == 2
    assert opt.param_groups[0]["params"][0] is a.icv_encoder.alpha            # "alpha" group first, with alpha_lr
    tr = ICVTrainer(b, total_steps=20, accumulate_grad_batches=1, grad_clip=None)
    batch = dict(zip(("query_inputs", "inputs", "query_x_length", "in_context_length"), _args(z)))
    for step in range(3):
        opt.zero_grad()
        loss = a.training_step(dict(batch), step)
        loss.backward()
        opt.step()
        sched.step()
        log = tr.micro_batch(*_args(z))
        assert abs(float(a.logged["kl_loss"]) - log["kl_loss"]) <= 2e-2 * abs(log["kl_loss"]) + 1e-6
        assert "alpha/alpha-0" in a.logged and "temperature" in a.logged
    for name in ("icv", "alpha"):
        pa, pb = getattr(a.icv_encoder, name).detach(), getattr(b.icv_encoder, name).detach()
        assert torch.allclose(pa, pb, rtol=0, atol=1e-7), f"{name} diverged: {(pa - pb).abs().max():.3e}"
    assert not torch.equal(a.icv_encoder.icv.detach().cpu(), T(z["enc_icv"])), "the optimiser must have moved the parameters"


def test_apply_icv_intervention_edit_function_tensor_tuple_and_passthrough():
    """ref:icv_src/icv_model/icv_intervention.py:61-86 — the public edit-function factory."""
    from icv_src.icv_model.icv_intervention import LearnableICVInterventionLMM
    w = LearnableICVInterventionLMM(torch.nn.Identity(), True, [3, 7], "model.layers.<LAYER_NUM>", 8)
    g = torch.Generator().manual_seed(11)
    h = torch.randn(2, 5, 64, generator=g)
    icv = torch.randn(1, 2, 64, generator=g) * 0.1
    fn = w.apply_icv_intervention(w.intervention_layer_names, icv.to(DEV))
    want = O.inject_renorm(h, icv[:, 1].unsqueeze(1))
    got = fn(h.to(DEV), "model.layers.7")
    assert (got.cpu() - want).abs().max() <= 1e-5 * want.abs().max()
    tup = fn((h.to(DEV), "cache", 3), "model.layers.7")
    assert isinstance(tup, tuple) and tup[1:] == ("cache", 3) and torch.equal(tup[0], got)
    same = fn(h.to(DEV), "model.layers.5")                       # not an edited layer: returned untouched
    assert torch.equal(same.cpu(), h)
    got3 = fn(h.to(DEV), "model.layers.3")
    assert (got3.cpu() - O.inject_renorm(h, icv[:, 0].unsqueeze(1))).abs().max() <= 1e-5 * want.abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_learnable_temperature_gets_its_gradient(dtype):
    """learnable_t (ref:icv_src/icv_module.py:49-52): `temperature` is a Parameter with requires_grad, and the reference's
    autograd differentiates  T^2 * mean_rows sum_v p (log(p+eps) - log(q+eps)),  p = softmax(tea/T), q = softmax(stu/T), with
    respect to it (ref :121-134).  Native: MaskedKLFn + licv_kl_rows_dtemp, against torch autograd of the same formula in fp64
    on the same (dtype-rounded) logits; the student-logit gradient must be unchanged by the extra output."""
    from licv.autograd import MaskedKLFn
    g = torch.Generator().manual_seed(7)
    R, V, n = 12, 1000, 5
    stu = (torch.randn(R, V, generator=g) * 2.0).to(dtype).to(DEV)
    tea = (torch.randn(R, V, generator=g) * 2.0).to(dtype).to(DEV)
    rows_s = torch.tensor([1, 3, 4, 8, 11], device=DEV)
    rows_t = torch.tensor([0, 2, 5, 9, 10], device=DEV)
    for temp in (1.0, 1.7):
        t_param = torch.nn.Parameter(torch.tensor(temp, device=DEV))
        s_in = stu.clone().requires_grad_(True)
        loss = MaskedKLFn.apply(s_in, tea, rows_s, rows_t, temp, 1e-6, t_param)
        loss.backward()
        assert t_param.grad is not None and t_param.grad.shape == t_param.shape
        # reference formula, fp64 autograd
        T64 = torch.tensor(temp, dtype=torch.float64, requires_grad=True)
        s64 = stu[rows_s].double().cpu() / T64
        t64 = tea[rows_t].double().cpu() / T64
        p, q = torch.softmax(t64, -1), torch.softmax(s64, -1)
        ref = (p * ((p + 1e-6).log() - (q + 1e-6).log())).sum(-1).mean() * T64 ** 2
        ref.backward()
        want = float(T64.grad)
        got = float(t_param.grad)
        tol = (2e-2 if dtype == torch.bfloat16 else 2e-4) * max(abs(want), 1e-3) + (5e-3 if dtype == torch.bfloat16 else 1e-5)
        assert abs(got - want) <= tol, (temp, dtype, got, want)
        # without the Parameter the node is not differentiable in T and the logit gradient is bit-identical
        s2 = stu.clone().requires_grad_(True)
        MaskedKLFn.apply(s2, tea, rows_s, rows_t, temp, 1e-6).backward()
        assert torch.equal(s2.grad, s_in.grad)
    assert n == rows_s.numel()


def test_module_with_learnable_t_fills_temperature_grad_and_caches_the_host_value(golden):
    z = golden("g6_loss")
    mod = _module(temp=2.0)
    mod.temperature.requires_grad_(True)
    _load(mod, z)
    loss_dict, _ = mod(*_args(z))
    loss_dict["loss"].backward()
    assert mod.temperature.grad is not None and torch.isfinite(mod.temperature.grad).all()
    assert mod.icv_encoder.icv.grad is not None
    # the host copy of T is read once per value of the Parameter, not once per call
    v0 = mod._t_host
    mod(*_args(z))
    assert mod._t_host is v0
    with torch.no_grad():
        mod.temperature.mul_(0.5)
    assert mod._temperature_value() == 1.0


def test_calculate_kl_divergence_is_differentiable_in_temperature_like_the_reference_formula(golden):
    """The public method (ref:icv_src/icv_module.py:121-134) and the forward()'s row-index form give the same temperature gradient
    with learnable_t; without grad recording the method stays the plain kernel call."""
    from oracle import icv_ref as O
    mod = _module(temp=2.0)
    mod.temperature.requires_grad_(True)
    g_ = torch.Generator().manual_seed(3)
    stu = (torch.randn(11, 98, generator=g_) * 2).to(DEV)
    tea = (torch.randn(11, 98, generator=g_) * 2).to(DEV)
    s_in = stu.clone().requires_grad_(True)
    kl = mod.calculate_kl_divergence(s_in, tea)
    kl.backward()
    t64 = torch.tensor(2.0, dtype=torch.float64, requires_grad=True)
    s64 = stu.double().cpu().requires_grad_(True)
    ref = O.kl_divergence(s64, tea.double().cpu(), t64, 1e-6)
    ref.backward()
    assert abs(float(kl) - float(ref)) <= 1e-4 * abs(float(ref))
    assert abs(float(mod.temperature.grad) - float(t64.grad)) <= 2e-3 * abs(float(t64.grad)) + 1e-6
    assert (s_in.grad.float().cpu() - s64.grad.float()).abs().max() <= 2 ** -7 * s64.grad.abs().max()
    with torch.no_grad():
        assert abs(float(mod.calculate_kl_divergence(stu, tea)) - float(ref)) <= 1e-4 * abs(float(ref))
