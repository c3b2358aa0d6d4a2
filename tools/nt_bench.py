#!/usr/bin/env python3
"""Non-temporal weight loads (licv_gemm_experiment knob 11: bit 0 = the 128-tile mid kernel's W pieces, bit 1 = the skinny kernel's weight
stream) on the weight-streaming shapes, COLD (as tools/stream_bench.py: > 600 MB of distinct matrices cycled), default policy beside it.
MI355X_MICROARCH.md, price list row nt-weights: once-read bytes land ~18 % sooner with nt; a replay from a warm cache loses."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

SHAPES = [(256, 12288, 4096), (256, 4096, 4096), (256, 22016, 4096), (256, 4096, 11008), (256, 32002, 4096),
          (24, 12288, 4096), (24, 4096, 4096), (24, 22016, 4096), (24, 4096, 11008), (24, 32002, 4096)]
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K) in SHAPES:
    nbuf = max(2, -(-640 * 2 ** 20 // (N * K * 2)))
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    res, outs = {}, {}
    for knob in (0, 3):
        lib.licv_gemm_experiment(11, knob)
        for w in ws: ops.linear(a, w)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for w in ws: ops.linear(a, w)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / nbuf * 1e3)
        res[knob] = best
        outs[knob] = ops.linear(a, ws[0]).clone()
    lib.licv_gemm_experiment(11, 0)
    same = torch.equal(outs[0], outs[3])
    print(f"{M:4d} {N:6d} {K:6d}  default {res[0]:7.1f} us ({N * K * 2 / res[0] / 1e6:5.2f} TB/s)   nt {res[3]:7.1f} us ({N * K * 2 / res[3] / 1e6:5.2f} TB/s)   "
          f"{100 * (res[0] / res[3] - 1):+5.1f} %   results {'identical' if same else 'DIFFER'}", flush=True)
    del ws
