#!/usr/bin/env python3
"""SigLIP fc1 at 264 images x 972 patches (256608 x 4352 x 1152: 2.2 GB of output, run as two row blocks because it is past the
32-bit offsets of the 256-tile kernels): the 4-wave kernels against the 8-wave / staged ones this one projection used to fall back
to, fp8 and bf16."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops
lib = _lib.lib()
M, N, K = 256608, 4352, 1152
g = torch.Generator(device="cuda").manual_seed(5)
a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
aq, asc = ops.quantize_fp8(a); wq, wsc = ops.quantize_fp8(w)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for knob in (1, 0):
    lib.licv_gemm_experiment(8, knob)
    print("fp8 fc1 gelu_tanh, 4-wave kernel" if knob else "fp8 fc1 gelu_tanh, 8-wave kernel", f"{t(lambda: ops.linear_fp8(aq, asc, wq, wsc, bias=bias, act='gelu_tanh', out=out)):.0f} us")
lib.licv_gemm_experiment(8, 1)
for sel in (0, 40):
    lib.licv_gemm_select(sel)
    print("bf16 fc1 gelu_tanh, default" if sel == 0 else "bf16 fc1 gelu_tanh, quad64 staged", f"{t(lambda: ops.linear(a, w, bias=bias, act='gelu_tanh', out=out)):.0f} us")
lib.licv_gemm_select(0)
