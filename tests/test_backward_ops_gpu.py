"""Op-level parity of every BACKWARD kernel of the student pass (csrc/backward.hip, licv_ce_rows) against torch autograd of the
oracle's restatement of the forward, at the real widths of BASELINE configs[2] / configs[4]: H = 4096, per-head 128, I = 11008 /
14336, V = 32002 / 32003, 32 query heads over 8 kv heads.

The training signal of the reference is torch autograd through the frozen LMM (ref:icv_src/icv_module.py:97-98,108-118).  Autograd
in the reference's bf16 regime has well-defined rounding points (a bf16 product's gradient is a bf16 product); the kernels mirror
those points and do everything between them in fp32.  Bars, written per test:
  * fp32-in / fp32-out kernels: <= 2e-5 of the tensor scale against CPU autograd (fp32 accumulation order is the only freedom);
  * bf16 outputs: within one (two where two roundings stack) bf16 ulp of CPU autograd in the same dtypes, >= 98 % bit-identical;
  * derivatives with no torch rounding points to mirror (KL, CE): within one bf16 ulp of fp64 autograd on the same logits.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import icv_ref as O
from oracle import idefics2_ref as R2
from oracle import idefics_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ops():
    from licv import ops as _ops
    return _ops


def g(seed):
    return torch.Generator().manual_seed(seed)


def close_bf16(got, ref, ulps=1.0, exact_frac=0.98, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    floor = ref.abs().max() * 2 ** -8
    tol = ulps * (ref.abs() * 2 ** -7 + floor * 2 ** -1) * 1.001
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} elements beyond {ulps} bf16 ulp; max diff {(got - ref).abs().max():.3e}"
    frac = float((got == ref).float().mean())
    if exact_frac:
        assert frac >= exact_frac, f"{what}: only {frac:.4f} of elements bit-exact"
    return frac


# ------------------------------------------------------------------------------------------- RMSNorm
def _rms(flavour):
    # flavour 0: hf:idefics/modeling_idefics.py:342-350 (cast, then weight); flavour 1: hf:mistral/modeling_mistral.py:182-199
    return R.rms_norm if flavour == 0 else R2.rms_norm


@pytest.mark.parametrize("flavour", [0, 1], ids=["idefics", "mistral"])
@pytest.mark.parametrize("x_dt", [torch.float32, torch.bfloat16], ids=["stream_f32", "stream_bf16"])
@pytest.mark.parametrize("accumulate", [False, True])
def test_rmsnorm_bwd_hidden_4096_vs_autograd(flavour, x_dt, accumulate):
    """dx of the stream norms (input_layernorm / post_attention_layernorm / final norm): x is the residual stream (fp32 after the
    first hook, bf16 before), dy the gradient of the following projection's bf16 input, dx accumulates into the fp32 stream
    gradient.  Autograd's rounding points: the Idefics norm casts xhat to the weight's bf16 before the product, so
    d(w * x_bf16)/d x_bf16 = bf16(dy * w); the Mistral norm multiplies w into xhat.to(input_dtype): a bf16 product on a bf16
    stream, an fp32 product (fp32 output, cast by the projection's autocast) on the fp32 stream behind a hook."""
    rows, dim, eps = 8 * 32, 4096, 1e-6
    x = (torch.randn(rows, dim, generator=g(1)) * 3).to(x_dt)
    w = (1 + 0.1 * torch.randn(dim, generator=g(2))).to(torch.bfloat16)
    dy = (torch.randn(rows, dim, generator=g(3)) * 0.02).to(torch.bfloat16)
    acc0 = torch.randn(rows, dim, generator=g(4)) * 0.02
    xr = x.clone().requires_grad_(True)
    y = _rms(flavour)(xr, w, eps)
    assert y.dtype == (torch.float32 if (flavour == 1 and x_dt == torch.float32) else torch.bfloat16)
    y.to(torch.bfloat16).backward(dy)                  # (the cast is the projection's autocast; identity where y is bf16 already)
    ref = xr.grad.float() + (acc0 if accumulate else 0)
    dx = acc0.clone().to(DEV) if accumulate else torch.empty(rows, dim, device=DEV)
    ops().rmsnorm_bwd(x.to(DEV), w.to(DEV), dy.to(DEV), dx, eps, accumulate=accumulate, flavour=flavour)
    err = float((dx.cpu() - ref).abs().max())
    scale = float(ref.abs().max())
    if x_dt == torch.float32:
        assert err <= 2e-5 * scale, f"rmsnorm_bwd fp32 stream: {err:.3e} vs scale {scale:.3e}"
    else:
        # autograd returns the bf16 stream's gradient as bf16 (two bf16 contributions summed); the kernel keeps fp32:
        # held to one bf16 ulp of autograd's value
        tol = (xr.grad.float().abs() * 2 ** -7 + xr.grad.float().abs().max() * 2 ** -9) * 1.001
        assert bool(((dx.cpu() - ref).abs() <= tol).all()), f"rmsnorm_bwd bf16 stream: max diff {err:.3e}"


def test_rmsnorm_bwd_per_head_128_strided_vs_autograd():
    """The gated cross-attention's q RMSNorm over head_dim (hf:idefics/modeling_idefics.py:598-600), backward in the fused row
    layout the student pass uses: (tokens, heads * 128) bf16 in, bf16 out, rows addressed by (inner, ld).  Two references:
    bf16 autograd (which rounds the two paths into x - through the statistics and through the product - to bf16 separately and
    sums them: the kernel, which rounds once, must stay within two bf16 ulp of it) and the closed form in fp64 with autograd's
    one real rounding point g = bf16(dy * w) (the kernel's bf16 output within one ulp, >= 98 % bit-identical to its rounding)."""
    T_, nh, hd, eps = 8 * 32, 32, 128, 1e-6
    q = torch.randn(T_, nh * hd, generator=g(5)).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(hd, generator=g(6))).to(torch.bfloat16)
    dq = (torch.randn(T_, nh * hd, generator=g(7)) * 0.05).to(torch.bfloat16)
    qr = q.clone().requires_grad_(True)
    R.rms_norm(qr.view(T_, nh, hd), w, eps).backward(dq.view(T_, nh, hd))
    out = torch.empty(T_, nh * hd, dtype=torch.bfloat16, device=DEV)
    ops().rmsnorm_bwd(q.to(DEV), w.to(DEV), dq.to(DEV), out, eps, accumulate=False, inner=nh, ld_x=nh * hd, ld_dy=nh * hd,
                      ld_dx=nh * hd, rows=T_ * nh, dim=hd)
    close_bf16(out, qr.grad, ulps=2.0, exact_frac=0.0, what="per-head rmsnorm_bwd vs bf16 autograd")
    x64 = q.double().view(T_, nh, hd)
    g64 = (dq.view(T_, nh, hd) * w).double()                                   # bf16 product, as autograd forms it
    rs = torch.rsqrt(x64.pow(2).mean(-1, keepdim=True) + eps)
    xhat = x64 * rs
    closed = (rs * (g64 - xhat * (g64 * xhat).mean(-1, keepdim=True))).reshape(T_, nh * hd)
    close_bf16(out, closed.float().to(torch.bfloat16), ulps=1.0, exact_frac=0.98, what="per-head rmsnorm_bwd vs fp64 closed form")


# ------------------------------------------------------------------------------------------- SwiGLU
@pytest.mark.parametrize("inter", [11008, 14336], ids=["idefics9b", "idefics2_8b"])
def test_swiglu_bwd_full_width_vs_autograd(inter):
    """act = silu(gate) * up in bf16 (hf:idefics/modeling_idefics.py:445-446, hf:mistral/modeling_mistral.py:35-48) on the unfused
    (rows, 2I) [gate | up] buffer the student pass keeps; d gate / d up against bf16 autograd on the CPU."""
    rows = 8 * 32
    gu = torch.randn(rows, 2 * inter, generator=g(8)).to(torch.bfloat16)
    dact = (torch.randn(rows, inter, generator=g(9)) * 0.05).to(torch.bfloat16)
    gr = gu.clone().requires_grad_(True)
    act = F.silu(gr[:, :inter]) * gr[:, inter:]
    act.backward(dact)
    fwd = ops().swiglu(gu.to(DEV))
    assert torch.equal(fwd.cpu(), act.detach()) or close_bf16(fwd, act.detach(), what="swiglu fwd") >= 0.99
    got = ops().swiglu_bwd(gu.to(DEV), dact.to(DEV))
    close_bf16(got[:, inter:], gr.grad[:, inter:], ulps=1.0, exact_frac=0.99, what="d up")
    close_bf16(got[:, :inter], gr.grad[:, :inter], ulps=1.0, exact_frac=0.97, what="d gate")


# ------------------------------------------------------------------------------------------- residual branch
@pytest.mark.parametrize("scaled", [False, True], ids=["decoder_branch", "gated_cross_attention_branch"])
def test_branch_grad_vs_autograd(scaled):
    """h(fp32 stream) = residual + [tanh(alpha) *] x(bf16), rows with cross_attention_gate == 0 filled with zero first
    (hf:idefics/modeling_idefics.py:792-802): d x = bf16(bf16(dh) * tanh(alpha)), zero on gated rows — bit-exact."""
    rows, dim = 8 * 32, 4096
    dh = torch.randn(rows, dim, generator=g(10)) * 0.03
    x = torch.randn(rows, dim, generator=g(11)).to(torch.bfloat16).requires_grad_(True)
    res = torch.randn(rows, dim, generator=g(12))
    if scaled:
        t = torch.tanh(torch.tensor(0.37)).to(torch.bfloat16)
        gate = (torch.rand(rows, generator=g(13)) > 0.3).float()
        xm = x.masked_fill((gate == 0)[:, None], 0.0)
        h = res + t * xm
    else:
        t, gate = None, None
        h = res + x
    assert h.dtype == torch.float32
    h.backward(dh)
    got = ops().branch_grad(dh.to(DEV), scale=float(t) if scaled else None, row_gate=gate.to(DEV) if scaled else None)
    assert torch.equal(got.cpu(), x.grad), f"branch_grad differs on {int((got.cpu() != x.grad).sum())} elements"


# ------------------------------------------------------------------------------------------- GQA group sum
def test_head_group_sum_32q_8kv_vs_autograd_of_repeat_kv():
    """Backward of repeat_kv (hf:mistral/modeling_mistral.py, 32 query heads over 8 kv heads x 128): the per-query-head dK / dV the
    attention backward writes, summed over each group of 4 — against bf16 autograd of the oracle's expand + reshape."""
    rows, nkv, rep, hd = 8 * 32, 8, 4, 128
    d_heads = torch.randn(rows, nkv * rep * hd, generator=g(14)).to(torch.bfloat16)
    kv = torch.randn(1, nkv, rows, hd, generator=g(15)).to(torch.bfloat16).requires_grad_(True)
    rk = R2.repeat_kv(kv, rep)                                               # (1, 32, rows, 128)
    rk.backward(d_heads.view(rows, nkv * rep, hd).transpose(0, 1)[None])
    ref = kv.grad[0].transpose(0, 1).reshape(rows, nkv * hd)
    ld_out = (nkv * rep + 2 * nkv) * hd                                      # written in place into the fused dqkv row, at the K offset
    out = torch.zeros(rows, ld_out, dtype=torch.bfloat16, device=DEV)
    ops().head_group_sum(d_heads.to(DEV), out.view(-1)[nkv * rep * hd:], rows, nkv, rep, hd, nkv * rep * hd, ld_out)
    close_bf16(out[:, nkv * rep * hd: nkv * rep * hd + nkv * hd], ref, ulps=1.0, exact_frac=0.995, what="head_group_sum")
    assert float(out[:, : nkv * rep * hd].abs().max()) == 0 and float(out[:, nkv * rep * hd + nkv * hd:].abs().max()) == 0


# ------------------------------------------------------------------------------------------- KL rows
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("temp", [1.0, 2.0])
@pytest.mark.parametrize("vocab", [32002, 32003])
def test_kl_rows_bwd_full_vocab_vs_autograd(dt, temp, vocab):
    """d/d student-logits of ref:icv_src/icv_module.py:121-134 on the masked rows, V = 32002 (Idefics) / 32003 (Idefics2), rows
    picked by index out of padded (B*S, ld) buffers.  Reference: fp64 autograd of the oracle's formula on the same logit values;
    the kernel's bf16 output within one bf16 ulp of it."""
    n_all, n = 8 * 32, 29
    ld = (vocab + 7) // 8 * 8
    stu = torch.zeros(n_all, ld)
    tea = torch.zeros(n_all + 5, ld)
    stu[:, :vocab] = torch.randn(n_all, vocab, generator=g(16)) * 3
    tea[:, :vocab] = torch.randn(n_all + 5, vocab, generator=g(17)) * 3
    stu, tea = stu.to(dt), tea.to(dt)
    srows = torch.randperm(n_all, generator=g(18))[:n].sort().values
    trows = torch.randperm(n_all + 5, generator=g(19))[:n].sort().values
    s64 = stu[srows][:, :vocab].double().requires_grad_(True)
    kl = O.kl_divergence(s64, tea[trows][:, :vocab].double(), temp, 1e-6)
    kl.backward()
    up = 0.5                                                                      # upstream d loss / d kl (grad accumulation)
    got = ops().kl_rows_bwd(stu.to(DEV), tea.to(DEV), srows.to(DEV), trows.to(DEV), vocab, temp, 1e-6, upstream=up)
    assert got.shape == (n, ld) and float(got[:, vocab:].abs().max()) == 0       # pad columns feed the head's dgrad GEMM as zeros
    close_bf16(got[:, :vocab], (up * s64.grad).float().to(torch.bfloat16), ulps=1.0, exact_frac=0.90, what="kl_rows_bwd")
    # the forward value on the same rows (already pinned in test_ops_gpu.py at small V) for completeness at full V
    f = ops().kl_rows(stu.to(DEV), tea.to(DEV), srows.to(DEV), trows.to(DEV), vocab, temp, 1e-6).mean() * temp ** 2
    same_dt = O.kl_divergence(stu[srows][:, :vocab], tea[trows][:, :vocab], temp, 1e-6)
    tol = 1e-4 if dt == torch.float32 else 3e-2
    assert abs(float(f) - float(same_dt)) <= tol * abs(float(same_dt)) + 1e-6, (float(f), float(same_dt), float(kl))


# ------------------------------------------------------------------------------------------- CE rows
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_ce_rows_full_vocab_loss_and_gradient_vs_autograd(dt):
    """The "hard" loss (ref:icv_src/icv_module.py:94-95,111-117 -> HF ForCausalLMLoss: logits upcast to fp32, mean over kept
    positions) at V = 32002: per-row loss against F.cross_entropy, gradient against autograd, plain / accumulate / scattered."""
    vocab, n_all, n = 32002, 8 * 32, 37
    ld = 32008
    lg = torch.zeros(n_all, ld)
    lg[:, :vocab] = torch.randn(n_all, vocab, generator=g(20)) * 4
    lg = lg.to(dt)
    rows = torch.randperm(n_all, generator=g(21))[:n].sort().values
    labels = torch.randint(0, vocab, (n,), generator=g(22))
    labels[0], labels[1] = vocab - 1, 0
    z = lg[rows][:, :vocab].double().requires_grad_(True)
    per_row = F.cross_entropy(z, labels, reduction="none")
    coef = 0.5 / n                                                                # hard_loss_weight / kept positions
    (per_row.sum() * coef).backward()
    grad = torch.zeros(n, ld, dtype=torch.bfloat16, device=DEV)
    loss = ops().ce_rows(lg.to(DEV), rows.to(DEV), labels.to(DEV), vocab, grad=grad, grad_coef=coef)
    assert float((loss.cpu().double() - per_row.detach()).abs().max()) <= 2e-5 * float(per_row.max())
    close_bf16(grad[:, :vocab], z.grad.float().to(torch.bfloat16), ulps=1.0, exact_frac=0.90, what="ce grad")
    assert float(grad[:, vocab:].abs().max()) == 0
    # accumulate on top of an existing gradient (KL rows first, then the CE term), rows scattered by grad_rows, coefficient on the device
    base = (torch.randn(n + 3, ld, generator=g(23)) * 1e-4).to(torch.bfloat16)
    base[:, vocab:] = 0
    dst = torch.randperm(n + 3, generator=g(24))[:n]
    acc = base.clone().to(DEV)
    ops().ce_rows(lg.to(DEV), rows.to(DEV), labels.to(DEV), vocab, grad=acc, grad_coef=1.0, grad_rows=dst.to(DEV), accumulate=True,
                  want_loss=False, grad_coef_dev=torch.tensor([coef], device=DEV))
    ref = base.float()
    ref[dst, :vocab] += z.grad.float()
    close_bf16(acc[:, :vocab], ref[:, :vocab].to(torch.bfloat16), ulps=1.0, exact_frac=0.90, what="ce grad accumulate")


# ------------------------------------------------------------------------------------------- attention at real head geometry
@pytest.mark.parametrize("nkv", [32, 8], ids=["mha_32x128", "gqa_32q_8kv_x128"])
def test_attention_bwd_real_head_geometry_vs_autograd(nkv):
    """Causal self-attention backward at the 9B / 8B head geometry (32 heads x 128, B = 8, S = 32, right padding) against bf16
    autograd through the oracle's eager attention (hf:idefics/modeling_idefics.py:450-470; GQA: repeat_kv) — dQ, dK, dV with the
    GQA group reduction applied as the student pass applies it."""
    B, S, nh, hd = 8, 32, 32, 128
    rep = nh // nkv
    qd, kd = nh * hd, nkv * hd
    ldq = qd + 2 * kd
    qkv = torch.randn(B * S, ldq, generator=g(25)).to(torch.bfloat16)
    dout = (torch.randn(B * S, qd, generator=g(26)) * 0.05).to(torch.bfloat16)
    key_valid = torch.ones(B, S, dtype=torch.int32)
    for b in range(B):
        key_valid[b, S - b:] = 0 if b else 1                                      # right padding of 0..7 tokens
    x = qkv.clone().requires_grad_(True)
    q = x[:, :qd].view(B, S, nh, hd).transpose(1, 2)
    k = x[:, qd:qd + kd].view(B, S, nkv, hd).transpose(1, 2)
    v = x[:, qd + kd:].view(B, S, nkv, hd).transpose(1, 2)
    if rep > 1:
        k, v = R2.repeat_kv(k, rep), R2.repeat_kv(v, rep)
    allowed = (torch.arange(S)[None, :] <= torch.arange(S)[:, None])[None] & key_valid.bool()[:, None, :]
    mask = torch.zeros(B, 1, S, S).masked_fill(~allowed[:, None], torch.finfo(torch.float32).min).to(torch.bfloat16)
    o = R.eager_attention(q, k, v, mask, hd ** -0.5)                              # (B, S, nh, hd)
    o.reshape(B * S, qd).backward(dout)
    ref = x.grad
    o_ = ops()
    d = qkv.to(DEV)
    dqkv = torch.zeros_like(d)
    if rep == 1:
        o_.attention_bwd_small(d, d.view(-1)[qd:], d.view(-1)[qd + kd:], dout.to(DEV), B, S, S, nh, nkv, hd, S * ldq, ldq, S * ldq, ldq,
                               hd ** -0.5, 1, dqkv, S * ldq, ldq, dk=dqkv.view(-1)[qd:], dv=dqkv.view(-1)[qd + kd:], dkv_bs=S * ldq,
                               dkv_rs=ldq, key_valid=key_valid.to(DEV))
    else:
        dkv_heads = torch.empty((B * S, 2 * qd), dtype=torch.bfloat16, device=DEV)
        o_.attention_bwd_small(d, d.view(-1)[qd:], d.view(-1)[qd + kd:], dout.to(DEV), B, S, S, nh, nkv, hd, S * ldq, ldq, S * ldq, ldq,
                               hd ** -0.5, 1, dqkv, S * ldq, ldq, dk=dkv_heads, dv=dkv_heads.view(-1)[qd:], dkv_bs=S * 2 * qd,
                               dkv_rs=2 * qd, key_valid=key_valid.to(DEV))
        o_.head_group_sum(dkv_heads, dqkv.view(-1)[qd:], B * S, nkv, rep, hd, 2 * qd, ldq)
        o_.head_group_sum(dkv_heads.view(-1)[qd:], dqkv.view(-1)[qd + kd:], B * S, nkv, rep, hd, 2 * qd, ldq)
    # rows past a sample's length receive a gradient in HF too (their queries attend to the valid keys); compare everything
    for name, sl in (("dQ", slice(0, qd)), ("dK", slice(qd, qd + kd)), ("dV", slice(qd + kd, ldq))):
        r_, g_ = ref[:, sl].float(), dqkv[:, sl].float().cpu()
        tol = 3 * 2.0 ** -8 * float(r_.abs().max())
        err = float((g_ - r_).abs().max())
        assert err <= tol, f"{name}: {err:.3e} > {tol:.3e}"
        cos = F.cosine_similarity(g_.reshape(1, -1), r_.reshape(1, -1)).item()
        assert cos > 0.9999, f"{name}: cosine {cos}"


# ------------------------------------------------------------------------------------------- round-4 forms of two backward kernels
@pytest.mark.parametrize("shape", [(8, 32, 32, 32, 8, 128, 1), (2, 32, 64, 4, 4, 64, 2), (3, 5, 7, 6, 2, 80, 1), (1, 100, 100, 2, 2, 128, 1)],
                         ids=["student_gqa", "perceiver_like_unmasked", "odd_sizes", "operands_do_not_fit_lds"])
def test_attention_bwd_staged_operands_equal_global_operands_bitwise(shape):
    """licv_attn_bwd_small with the head's Q / K / V / dO rows staged in LDS (licv_backward_option 0, the default where they fit) against
    the same kernel reading them from global memory: same arithmetic in the same order, identical bits - causal + key padding, no
    mask, odd sizes (a pair count that is not a multiple of 4, head dim 80), and a shape whose operands do not fit (falls back)."""
    from licv import _lib
    B, Sq, Sk, nh, nkv, hd, mode = shape
    qd, kd = nh * hd, nkv * hd
    q = torch.randn(B * Sq, qd, generator=g(41)).to(torch.bfloat16).to(DEV)
    kv = torch.randn(B * Sk, 2 * kd, generator=g(42)).to(torch.bfloat16).to(DEV)
    dout = (torch.randn(B * Sq, qd, generator=g(43)) * 0.05).to(torch.bfloat16).to(DEV)
    key_valid = (torch.rand(B, Sk, generator=g(44)) > 0.2).to(torch.int32)
    key_valid[:, 0] = 1
    outs = []
    try:
        for staged in (0, 1):
            _lib.check(_lib.lib().licv_backward_option(0, staged))
            dq = torch.zeros_like(q)
            dkv = torch.zeros((B * Sk, 2 * qd), dtype=torch.bfloat16, device=DEV)
            ops().attention_bwd_small(q, kv, kv.view(-1)[kd:], dout, B, Sq, Sk, nh, nkv, hd, Sq * qd, qd, Sk * 2 * kd, 2 * kd, hd ** -0.5, mode,
                                      dq, Sq * qd, qd, dk=dkv, dv=dkv.view(-1)[qd:], dkv_bs=Sk * 2 * qd, dkv_rs=2 * qd,
                                      key_valid=key_valid.to(DEV))
            outs.append((dq, dkv))
    finally:
        _lib.check(_lib.lib().licv_backward_option(0, 1))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[1][0].float().abs().max()) > 0 and float(outs[1][1].float().abs().max()) > 0


@pytest.mark.parametrize("dim", [1024, 1152, 4096])
@pytest.mark.parametrize("x_dt", [torch.float32, torch.bfloat16])
def test_rmsnorm_bwd_four_waves_per_row_equal_one_wave_per_row_bitwise(dim, x_dt):
    """licv_rmsnorm_bwd on four waves per row (licv_backward_option 1, the default from 1024 elements on) against one wave per row:
    both form the two row statistics in four chunk groups added in group order - identical bits, with and without accumulation."""
    from licv import _lib
    rows = 77
    x = torch.randn(rows, dim, generator=g(45)).to(x_dt).to(DEV)
    w = (1 + 0.1 * torch.randn(dim, generator=g(46))).to(torch.bfloat16).to(DEV)
    dy = torch.randn(rows, dim, generator=g(47)).to(x_dt).to(DEV)
    base = torch.randn(rows, dim, generator=g(48)).to(x_dt).to(DEV)
    outs = []
    try:
        for wide in (0, 1):
            _lib.check(_lib.lib().licv_backward_option(1, wide))
            for acc in (False, True):
                dx = base.clone()
                ops().rmsnorm_bwd(x, w, dy, dx, 1e-6, accumulate=acc, flavour=1)
                outs.append(dx)
    finally:
        _lib.check(_lib.lib().licv_backward_option(1, 1))
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
    assert not torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("x_dt,flavour", [(torch.float32, 1), (torch.float32, 0), (torch.bfloat16, 0)])
@pytest.mark.parametrize("K", [11008, 22016, 12288])
def test_rmsnorm_bwd_fed_by_split_k_slices_equals_gemm_then_norm_bitwise(K, x_dt, flavour):
    """ops.rmsnorm_bwd_from: the dgrad GEMM's producer half + licv_rmsnorm_bwd_ws (the norm's backward sums the fp32 slices itself) against
    linear() + rmsnorm_bwd() - the three dgrad shapes of the 256-row student (d gu . [gate|up], d act . down at Idefics2's width, dqkv . qkv),
    both norm flavours, fp32 and bf16 streams, accumulating into a non-zero dx."""
    M, H = 256, 4096
    a = (torch.randn(M, K, generator=g(51)) * 0.05).to(torch.bfloat16).to(DEV)
    wt = (torch.randn(H, K, generator=g(52)) * K ** -0.5).to(torch.bfloat16).to(DEV)
    x = torch.randn(M, H, generator=g(53)).to(x_dt).to(DEV)
    w = (1 + 0.1 * torch.randn(H, generator=g(54))).to(torch.bfloat16).to(DEV)
    base = torch.randn(M, H, generator=g(55)).to(x_dt).to(DEV)
    o = ops()
    assert o.linear_produce(a, wt) is not None, "the plan must split this shape"
    for acc in (True, False):
        d1, d2 = base.clone(), base.clone()
        o.rmsnorm_bwd(x, w, o.linear(a, wt), d1, 1e-6, accumulate=acc, flavour=flavour)
        o.rmsnorm_bwd_from(x, w, a, wt, d2, 1e-6, accumulate=acc, flavour=flavour)
        assert torch.equal(d1, d2), (K, x_dt, flavour, acc)
