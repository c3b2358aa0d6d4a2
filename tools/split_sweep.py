#!/usr/bin/env python3
"""Forced split-K counts (licv_gemm_experiment knob 5; 0 = the plan's own choice) on the 128-tile route, cold buffers: the 8-image
vision tower's projections and two M = 256 language shapes - the data the plan's cost model is fitted to."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K, epi) in [(2056, 1280, 5120, "res"), (2056, 1280, 1280, "res"), (2056, 3840, 1280, "bias"), (2056, 5120, 1280, "gelu"), (256, 4096, 4096, ""), (256, 4096, 11008, "")]:
    nbuf = max(2, -(-300 * 2 ** 20 // ((N * K + M * K + M * N) * 2)))
    As = [torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16) for _ in range(nbuf)]
    Ws = [(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16) for _ in range(nbuf)]
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(res=dict(bias=bias, residual=res), bias=dict(bias=bias), gelu=dict(bias=bias, act="gelu")).get(epi, {})
    line = f"{M} {N} {K} {epi:5s}"
    for sp in (0, 1, 2, 3, 4, 6, 8):
        lib.licv_gemm_experiment(5, sp)
        try:
            for i in range(nbuf): ops.linear(As[i], Ws[i], out=out, **kw)
        except Exception as e:
            line += f"  sp{sp}: err"; continue
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nbuf): ops.linear(As[i], Ws[i], out=out, **kw)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / nbuf * 1e3)
        line += f"  sp{sp}: {best:5.1f}"
    lib.licv_gemm_experiment(5, 0)
    print(line, flush=True)
