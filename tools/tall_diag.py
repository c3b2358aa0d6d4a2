#!/usr/bin/env python3
"""Where the tall kernel's split-K output differs from the mid kernel's (diagnostic)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops
lib = _lib.lib()
g = torch.Generator().manual_seed(3)
M, N, K = 256, 4096, 4096
a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
for sp in (2, 4, 8, 16):
    lib.licv_gemm_experiment(5, sp)
    lib.licv_gemm_select(70)
    ref = ops.linear(a, w).clone()
    lib.licv_gemm_select(71)
    for rep in range(4):
        out = ops.linear(a, w).clone()
        bad = (out != ref)
        nb = int(bad.sum())
        if nb:
            idx = bad.nonzero()
            rows = sorted(set((idx[:, 0] // 16).tolist()))
            cols = sorted(set((idx[:, 1] // 16).tolist()))
            err = (out.float() - ref.float()).abs().max().item()
            print(f"sp {sp} rep {rep}: {nb} differ, max err {err:.4f}; row blocks {rows[:20]}{'...' if len(rows) > 20 else ''} ({len(rows)}), col blocks {cols[:20]}{'...' if len(cols) > 20 else ''} ({len(cols)})", flush=True)
        else:
            print(f"sp {sp} rep {rep}: identical", flush=True)
lib.licv_gemm_experiment(5, 0)
lib.licv_gemm_select(0)
