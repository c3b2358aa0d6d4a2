#!/usr/bin/env python3
"""Mid-M GEMM shapes (the 32-token student / generate prefill: M = 256; the vision tower on 8 images: M = 2056; decode: M = 24):
time per call and weight-stream rate through licv.ops.linear (split-K plan included), per licv_gemm_select value.
Usage: python tools/mid_bench.py [select ...]   (0 = default dispatch, 1 = register-staged 128-tile kernel, 70 = mid kernel)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

SHAPES = [(256, 12288, 4096, "plain"), (256, 4096, 4096, "res32"), (256, 22016, 4096, "swiglu"), (256, 4096, 11008, "res32"),
          (256, 32002, 4096, "plain"), (512, 8192, 1280, "plain"),
          (2056, 3840, 1280, "bias"), (2056, 1280, 1280, "bias+res"), (2056, 5120, 1280, "bias+gelu"), (2056, 1280, 5120, "bias+res"),
          (24, 12288, 4096, "plain"), (24, 4096, 11008, "res32"),
          (1376, 12288, 4096, "plain"), (1376, 4096, 4096, "res32"), (1376, 28672, 4096, "swiglu"), (1376, 4096, 14336, "res32"),
          (512, 4096, 1280, "plain"), (512, 5120, 1280, "plain"), (512, 1280, 5120, "plain")]
# "sweep": the 128-tile route with forced split counts (knob 5) next to the 256-tile kernels (60) and the plan's own choice (0)
SWEEP = len(sys.argv) > 1 and sys.argv[1] == "sweep"
sels = ([60, 0] + [700 + c for c in (1, 2, 3, 4, 6, 8, 12, 16)]) if SWEEP else ([int(x) for x in sys.argv[1:]] or [1, 0])
lib = _lib.lib()
_lib.lab()          # experiment kernels (csrc/lab/) register themselves with licv_gemm_select
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K, epi) in SHAPES:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    n_out = N // 2 if epi == "swiglu" else N
    kw = {}
    if "bias" in epi: kw["bias"] = (torch.randn(N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    if "gelu" in epi: kw["act"] = "gelu"
    if epi == "swiglu": kw["swiglu"] = True
    res32 = torch.randn(M, n_out, device="cuda", generator=g) if "res32" in epi else None
    res16 = torch.randn(M, n_out, device="cuda", generator=g).to(torch.bfloat16) if epi.endswith("+res") else None
    def run():
        if res32 is not None: return ops.linear(a, w, residual=res32, out=res32, **kw)
        if res16 is not None: return ops.linear(a, w, residual=res16, out=res16, **kw)
        return ops.linear(a, w, **kw)
    out, ref = {}, None
    for sel in sels:
        if sel >= 700:                                   # 128-tile route, forced split count
            if (sel - 700) > 1 and (K // 64) // (sel - 700) < 2: continue
            lib.licv_gemm_select(70); lib.licv_gemm_experiment(5, sel - 700); lib.licv_gemm_experiment(6, 1 << 30)
        else:
            lib.licv_gemm_select(sel); lib.licv_gemm_experiment(5, 0); lib.licv_gemm_experiment(6, 160)
        for _ in range(3): run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        out[sel] = best
    lib.licv_gemm_select(0); lib.licv_gemm_experiment(5, 0); lib.licv_gemm_experiment(6, 160)
    if SWEEP:
        print(f"{M:5d} {N:6d} {K:6d} {epi:10s} " + "  ".join(f"{('x%d' % (s - 700)) if s >= 700 else s}: {out[s]:6.1f}" for s in sels if s in out), flush=True)
    else:
        print(f"{M:5d} {N:6d} {K:6d} {epi:10s} " + "  ".join(f"{s}: {out[s]:7.1f} us ({N * K * 2 / out[s] / 1e6:5.2f} TB/s, {2.0 * M * N * K / out[s] / 1e6:6.0f} TF)" for s in sels), flush=True)
