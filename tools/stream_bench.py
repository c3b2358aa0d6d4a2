#!/usr/bin/env python3
"""Weight-streaming GEMM shapes COLD: every call reads a different weight matrix, the set cycled through is > 600 MB, so nothing is
served from the 256 MB Infinity Cache (tools/mid_bench.py re-reads one matrix - at 100 MB it stays on-die and the rate is inflated).
The student (M = 256) and the decode steps (M = 24) of the 9B run in exactly this regime.
Usage: python tools/stream_bench.py [select ...]  (0 = default dispatch; 'blas' = torch.matmul)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

SHAPES = [(256, 12288, 4096), (256, 4096, 4096), (256, 22016, 4096), (256, 4096, 11008), (256, 32002, 4096), (256, 8192, 1280),
          (24, 12288, 4096), (24, 4096, 4096), (24, 22016, 4096), (24, 4096, 11008), (24, 32002, 4096)]
sels = sys.argv[1:] or ["0", "blas"]
lib = _lib.lib()
_lib.lab()          # experiment kernels (csrc/lab/) register themselves with licv_gemm_select
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K) in SHAPES:
    nbuf = max(2, -(-640 * 2 ** 20 // (N * K * 2)))
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    res = {}
    for sel in sels:
        if sel == "blas":
            run = lambda w: torch.matmul(a, w.t())
        else:
            lib.licv_gemm_select(int(sel))
            run = lambda w: ops.linear(a, w)
        for w in ws: run(w)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for w in ws: run(w)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / nbuf * 1e3)
        res[sel] = best
    lib.licv_gemm_select(0)
    print(f"{M:4d} {N:6d} {K:6d}  W {N * K * 2 / 2**20:6.1f} MiB x{nbuf:3d}  " + "  ".join(f"{s}: {res[s]:7.1f} us ({N * K * 2 / res[s] / 1e6:5.2f} TB/s)" for s in sels), flush=True)
    del ws
