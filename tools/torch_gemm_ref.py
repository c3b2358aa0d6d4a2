#!/usr/bin/env python3
"""What PyTorch-ROCm's library GEMM (hipBLASLt / rocBLAS behind F.linear) reaches on the headline shapes, random bf16 operands:
a platform reference point for the hand-written kernels (measurement only, nothing of the product path calls it)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
import torch.nn.functional as F
from licv import ops

SHAPES = [(67848, 3840, 1280), (67848, 1280, 1280), (67848, 5120, 1280), (67848, 1280, 5120), (6400, 12288, 4096), (6400, 4096, 4096),
          (6400, 22016, 4096), (6400, 4096, 11008), (16896, 8192, 1280), (8192, 8192, 8192)]
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    res = {}
    for name, fn in (("torch", lambda: F.linear(a, w)), ("licv", lambda: ops.linear(a, w))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 0.0
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = max(best, 2.0 * M * N * K * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        res[name] = best
    print(f"{M:6d} {N:6d} {K:6d}  torch F.linear {res['torch']:7.1f} TF   licv {res['licv']:7.1f} TF", flush=True)
