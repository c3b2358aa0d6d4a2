"""fp8 path (BASELINE configs[4], "fp8 weights on CDNA4 MFMA").  The reference has no fp8 mode, so parity is defined in two layers:
  kernel level (exact):  the quantiser reproduces torch's e4m3fn cast byte for byte; the fp8 GEMM equals the fp64 product of
                         the DEQUANTISED operands within one bf16 rounding of the output (the only freedom is the order of
                         the fp32 accumulation);
  model level (derived): see tests/test_idefics2_gpu.py::test_idefics2_fp8_* — deviation from the bf16 engine.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
F8 = torch.float8_e4m3fn


def _ref_quant(x):
    sc = x.float().abs().amax(1).clamp_min(1e-12) / 448.0
    q = (x.float() / sc[:, None]).to(F8)
    return q.view(torch.uint8), sc


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,K", [(5, 64), (300, 1152), (129, 4096), (7, 14336), (33, 4352), (9, 16384), (3, 16392), (65, 520), (40, 1024)])
def test_quantize_rows_matches_torch_cast(dt, rows, K):
    from licv import ops
    g = torch.Generator().manual_seed(rows + K)
    x = (torch.randn(rows, K, generator=g) * torch.rand(rows, 1, generator=g) * 5).to(dt)
    x[0, :] = 0                                                   # an all-zero row: scale floor, zeros out
    x[1, 3] = 1000.0                                              # an outlier sets the row scale
    q, sc = ops.quantize_fp8(x.to(DEV))
    rq, rsc = _ref_quant(x)
    assert torch.equal(sc.cpu(), rsc)
    assert torch.equal(q.cpu(), rq)
    if K % 16 == 0:                                               # rows inside a wider buffer (the one-pass kernel reads past no row end)
        wide = torch.full((rows, K + 64), 1e4, dtype=dt)
        wide[:, :K] = x
        q2, sc2 = ops.quantize_fp8(wide.to(DEV)[:, :K])
        assert torch.equal(sc2.cpu(), rsc) and torch.equal(q2.cpu(), rq)


@pytest.mark.parametrize("M,N,K,epi", [(512, 256, 256, "none"), (1000, 3000, 1152, "bias_gelu"), (700, 512, 4096, "res16"),
                                       (640, 1024, 1024, "swiglu"), (513, 260, 2048, "res32"), (300, 100, 512, "none"),
                                       # the 4-wave kernel on the 128-deep MFMA (M >= 512, N % 128 == 0, K % 128 == 0, K >= 512): the shortest K it
                                       # takes, every epilogue family, a ragged last row tile, more tiles than CUs (the persistent seam)
                                       (512, 256, 512, "none"), (1000, 3072, 1152, "bias_gelu"), (1000, 1152, 4352, "bias"), (777, 1152, 1152, "res16"),
                                       (640, 1024, 1024, "swiglu128"), (4700, 4096, 1024, "bias"), (2056, 640, 5120, "none")])
def test_gemm_fp8_matches_dequantised_product(M, N, K, epi):
    from licv import ops
    if epi == "swiglu128":
        epi = "swiglu"
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    aq, asc = ops.quantize_fp8(a.to(DEV))
    wq, wsc = ops.quantize_fp8(w.to(DEV))
    ad = aq.cpu().view(F8).double() * asc.cpu().double()[:, None]
    wd = wq.cpu().view(F8).double() * wsc.cpu().double()[:, None]
    y = ad @ wd.T
    kw = {}
    if epi == "bias_gelu":
        bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
        y = torch.nn.functional.gelu((y + bias.double()).float().bfloat16().double())
        kw = dict(bias=bias.to(DEV), act="gelu")
    elif epi == "bias":
        bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
        y = y + bias.double()
        kw = dict(bias=bias.to(DEV))
    elif epi == "swiglu":
        kw = dict(swiglu=True)
    elif epi in ("res16", "res32"):
        rdt = torch.bfloat16 if epi == "res16" else torch.float32
        res = torch.randn(M, N, generator=g).to(rdt)
        y = y.float().bfloat16().double() + res.double()
        kw = dict(residual=res.to(DEV))
    if epi == "swiglu":
        wq2 = ops.pack_gate_up(w[: N // 2].to(DEV).contiguous(), w[N // 2:].to(DEV).contiguous())       # interleave, then quantise rows
        wq, wsc = ops.quantize_fp8(wq2)
        wd = wq.cpu().view(F8).double() * wsc.cpu().double()[:, None]
        yy = (ad @ wd.T).float().bfloat16().double()
        gate = torch.cat([yy[:, b:b + 16] for b in range(0, N, 32)], 1)
        up = torch.cat([yy[:, b + 16:b + 32] for b in range(0, N, 32)], 1)
        y = torch.nn.functional.silu(gate).float().bfloat16().double() * up
    out16 = ops.linear_fp8(aq, asc, wq, wsc, **kw)
    assert torch.equal(out16, ops.linear_fp8(aq, asc, wq, wsc, **kw))          # run to run
    out = out16.float().cpu().double()
    assert out.shape == y.shape
    tol = 2.0 ** -7 * y.abs().max()                               # one bf16 ulp at the output scale (+ fp32 accumulation order)
    assert (out - y).abs().max() <= tol, f"{float((out - y).abs().max()):.3e} > {float(tol):.3e}"
    # both fp8 kernels (8 waves on the 32-deep fp8 MFMA, 4 waves on the 128-deep one) multiply the same bytes: they may differ by the
    # order of the fp32 sums only
    from licv import _lib
    try:
        _lib.lib().licv_gemm_experiment(8, 0)
        other = ops.linear_fp8(aq, asc, wq, wsc, **kw).float().cpu().double()
    finally:
        _lib.lib().licv_gemm_experiment(8, 1)
    assert (other - y).abs().max() <= tol
    assert (other - out).abs().max() <= tol


@pytest.mark.parametrize("dim", [256, 1152, 4096])
def test_norm_kernels_write_the_fp8_rows_of_their_output(dim):
    """The producers of fp8 GEMM inputs (RMSNorm, residual add + RMSNorm, hook + RMSNorm, LayerNorm) write the e4m3 rows and their
    scales themselves: byte for byte what the row quantiser makes of the bf16 rows the plain kernels write (and the stream / hook
    outputs are untouched by the extra output)."""
    from licv import ops
    rows = 77
    g = torch.Generator().manual_seed(dim)
    w = (1 + 0.1 * torch.randn(dim, generator=g)).to(torch.bfloat16).to(DEV)
    b = (0.1 * torch.randn(dim, generator=g)).to(torch.bfloat16).to(DEV)
    for dt in (torch.bfloat16, torch.float32):
        for flavour in (0, 1):
            x = (torch.randn(rows, dim, generator=g) * 3).to(dt).to(DEV)
            x[3] = 0
            q, sc, out = ops.rmsnorm_q8(x, w, 1e-6, flavour, want_bf16=True)
            ref = ops.rmsnorm(x, w, 1e-6, flavour)
            rq, rs = ops.quantize_fp8(ref)
            assert torch.equal(out, ref) and torch.equal(q, rq) and torch.equal(sc, rs)
            q2, sc2 = ops.rmsnorm_q8(x, w, 1e-6, flavour)
            assert torch.equal(q2, rq) and torch.equal(sc2, rs)
            h1, h2 = x.clone(), x.clone()
            br = (torch.randn(rows, dim, generator=g)).to(torch.bfloat16).to(DEV)
            ref = ops.add_rmsnorm_(h1, br, w, 1e-6, flavour)
            q, sc = ops.add_rmsnorm_q8_(h2, br, w, 1e-6, flavour)
            rq, rs = ops.quantize_fp8(ref)
            assert torch.equal(h1, h2) and torch.equal(q, rq) and torch.equal(sc, rs)
        res = (torch.randn(rows, dim, generator=g)).to(dt).to(DEV)
        br = (torch.randn(rows, dim, generator=g)).to(torch.bfloat16).to(DEV)
        icv = (torch.randn(dim, generator=g) * 0.1).to(DEV)
        al = torch.tensor([0.7], device=DEV)
        o1, xn = ops.inject_renorm_add(br, icv, res, alpha=al, norm_weight=w, norm_eps=1e-6, norm_flavour=1)
        o2, q, sc = ops.inject_renorm_add_q8(br, icv, res, al, w, norm_eps=1e-6, norm_flavour=1)
        rq, rs = ops.quantize_fp8(xn)
        assert torch.equal(o1, o2) and torch.equal(q, rq) and torch.equal(sc, rs)
    if dim % 8 == 0:
        x = (torch.randn(rows, dim, generator=g) * 2 + 0.5).to(torch.bfloat16).to(DEV)
        ref = ops.layernorm(x, w, b, 1e-6)
        q, sc = ops.layernorm_q8(x, w, b, 1e-6)
        rq, rs = ops.quantize_fp8(ref)
        assert torch.equal(q, rq) and torch.equal(sc, rs)
