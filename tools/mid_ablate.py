#!/usr/bin/env python3
"""Timing-only ablations of the 128-tile kernel's operand stream at M = 256, cold (licv_gemm_experiment knob 9: 1 = no A pieces,
2 = no W pieces; the results of those runs are wrong by construction)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K) in [(256, 12288, 4096), (256, 22016, 4096), (256, 8192, 1280), (128, 12288, 4096)]:
    nbuf = max(2, -(-640 * 2 ** 20 // (N * K * 2)))
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    lib.licv_gemm_experiment(4, 0)                      # one pass (no split-K): the lone-workgroup regime
    res = {}
    for abl in (0, 1, 2):
        lib.licv_gemm_experiment(9, abl)
        for w in ws: ops.linear(a, w)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for w in ws: ops.linear(a, w)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / nbuf * 1e3)
        res[abl] = best
    lib.licv_gemm_experiment(9, 0); lib.licv_gemm_experiment(4, 1)
    print(f"{M} x {N} x {K}: full {res[0]:.1f} us, no A pieces {res[1]:.1f} us, no W pieces {res[2]:.1f} us", flush=True)
