#!/usr/bin/env python3
"""fp8 GEMM shapes of the Idefics2-8B 32-shot step (Mistral text stack at ~22 k token rows, SigLIP tower): both fp8 kernels and the bf16
kernel, TFLOP/s-equivalent (2 M N K / time)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops
lib = _lib.lib()
SHAPES = [(8192, 6144, 4096, "plain"), (8192, 4096, 4096, "plain"), (8192, 28672, 4096, "swiglu"), (8192, 4096, 14336, "plain"),
          (16384, 3456, 1152, "bias"), (16384, 4352, 1152, "bias_gelu"), (16384, 1152, 4352, "bias")]
g = torch.Generator(device="cuda").manual_seed(0)
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, epi) in SHAPES:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
    kw = {}
    if "bias" in epi: kw["bias"] = (torch.randn(N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    if "gelu" in epi: kw["act"] = "gelu"
    if epi == "swiglu": kw["swiglu"] = True
    aq, asc = ops.quantize_fp8(a)
    wq, wsc = ops.quantize_fp8(w)
    t16 = timed(lambda: ops.linear(a, w, **kw))
    lib.licv_gemm_experiment(8, 0)
    t8a = timed(lambda: ops.linear_fp8(aq, asc, wq, wsc, **kw))
    lib.licv_gemm_experiment(8, 1)
    t8b = timed(lambda: ops.linear_fp8(aq, asc, wq, wsc, **kw))
    tq = timed(lambda: ops.quantize_fp8(a))
    f = 2.0 * M * N * K / 1e6
    print(f"{M:6d} {N:6d} {K:6d} {epi:10s} bf16 {t16:7.1f} us {f / t16:6.0f} TF | fp8 32-deep {t8a:7.1f} us {f / t8a:6.0f} TF | fp8 128-deep {t8b:7.1f} us {f / t8b:6.0f} TF"
          f" ({t16 / t8b:4.2f}x bf16) | row quantiser {tq:6.1f} us", flush=True)
