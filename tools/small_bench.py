#!/usr/bin/env python3
"""Per-call device time of the row kernels at decode-step sizes (24 rows x 4096): where a launch-bound layer spends its microseconds."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import ops


def t(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows in (24, 256):
    H = 4096
    h32 = torch.randn(rows, H, device="cuda")
    h16 = h32.to(torch.bfloat16)
    w = torch.ones(H, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(H, device="cuda")
    out16 = torch.empty_like(h16)
    print(f"rows={rows}: rmsnorm f32 {t(lambda: ops.rmsnorm(h32, w, 1e-6, out=out16)):.1f} us, rmsnorm bf16 {t(lambda: ops.rmsnorm(h16, w, 1e-6, out=out16)):.1f} us, "
          f"inject+norm {t(lambda: ops.inject_renorm(h32, v, out=h32, norm_weight=w)):.1f} us, "
          f"empty-ish add {t(lambda: h32.add_(1.0)):.1f} us")
    a = torch.randn(rows, 4096, device="cuda").to(torch.bfloat16)
    for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)):
        wt = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        x = torch.randn(rows, K, device="cuda").to(torch.bfloat16)
        us = t(lambda: ops.linear(x, wt), 50)
        print(f"   linear {rows}x{N}x{K}: {us:.1f} us = {N * K * 2 / us / 1e6:.2f} TB/s of weights")
