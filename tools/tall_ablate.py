#!/usr/bin/env python3
"""Timing-only ablations of the tall kernel's operand streams (knob 9: 1 = no A pieces, 2 = no W loads, 3 = neither; results wrong):
a lone workgroup per CU (forced one pass / few splits), cold weights."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K, sp) in [(256, 4096, 4096, 1), (256, 4096, 4096, 8), (256, 22016, 4096, 1), (256, 32768, 4096, 1)]:
    nbuf = max(2, -(-640 * 2 ** 20 // (N * K * 2)))
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16) for _ in range(nbuf)]
    lib.licv_gemm_experiment(5, sp)
    line = f"{M} {N} {K} sp{sp}:"
    for sel, abls in ((71, (0, 1, 2, 3)), (70, (0, 1, 2))):
        lib.licv_gemm_select(sel)
        for abl in abls:
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
            line += f"  {'tall' if sel == 71 else 'mid'} abl{abl}: {best:6.1f}"
        lib.licv_gemm_experiment(9, 0)
    print(line, flush=True)
    del ws
lib.licv_gemm_experiment(5, 0); lib.licv_gemm_select(0)
