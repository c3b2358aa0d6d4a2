#!/usr/bin/env python3
"""One weight-streaming GEMM shape COLD (a ring of distinct weight matrices > 600 MB), a few rounds: the program rocprofv3 --pmc runs
for the wave-state / LDS / L2 counters of the M = 256 kernel.  argv: M N K [rounds] [gemm select]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

M, N, K = (int(v) for v in sys.argv[1:4])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
_lib.lib().licv_gemm_select(int(sys.argv[5]) if len(sys.argv) > 5 else 0)
import os
if os.environ.get("LICV_MID_DEPTH"):
    _lib.lib().licv_gemm_experiment(10, int(os.environ["LICV_MID_DEPTH"]))
if os.environ.get("LICV_ONE_PASS"):
    _lib.lib().licv_gemm_experiment(4, 0)
g = torch.Generator(device="cuda").manual_seed(1)
nbuf = max(2, -(-640 * 2 ** 20 // (N * K * 2)))
a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
for r in range(rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for w in ws:
        ops.linear(a, w)
    e1.record()
    torch.cuda.synchronize()
    print(f"{M} x {N} x {K} round {r}: {e0.elapsed_time(e1) / nbuf * 1e3:.1f} us per call ({nbuf} matrices)")
