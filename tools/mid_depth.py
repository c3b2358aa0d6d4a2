#!/usr/bin/env python3
"""The 128-tile mid kernel with two / four K tiles in flight (licv_gemm_experiment knob 10) on the M <= 256 shapes, cold: time per call
under the plan's split count and in one pass, outputs compared bit for bit."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)
SHAPES = [(256, 12288, 4096), (256, 4096, 4096), (256, 22016, 4096), (256, 4096, 11008), (256, 32002, 4096), (256, 8192, 1280), (128, 12288, 4096), (64, 4096, 4096)]
for (M, N, K) in SHAPES:
    nbuf = max(2, -(-640 * 2 ** 20 // (N * K * 2)))
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    line = f"{M:4d} {N:6d} {K:6d}"
    for onepass in (0, 1):
        lib.licv_gemm_experiment(4, 0 if onepass else 1)
        outs = {}
        for depth in (2, 4):
            lib.licv_gemm_experiment(10, depth)
            for w in ws: o = ops.linear(a, w)
            outs[depth] = ops.linear(a, ws[0]).clone()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for w in ws: ops.linear(a, w)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / nbuf * 1e3)
            line += f"  {'one pass' if onepass else 'plan'} D={depth}: {best:6.1f} us"
        line += " same" if torch.equal(outs[2], outs[4]) else " DIFF"
    lib.licv_gemm_experiment(10, 0); lib.licv_gemm_experiment(4, 1)
    print(line, flush=True)
    del ws
