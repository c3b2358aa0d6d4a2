#!/usr/bin/env python3
"""Tile-size routing check: shapes of the Idefics2 1-shot step (SigLIP at 16 x 972 patches, text at 8 x 172 tokens) and of small
vision batches under the default route, the 256-tile flow64 kernel forced (select 60) and the 128-tile mid kernel forced (70).
Buffers are cycled (> 300 MB of weights + activations per shape) so nothing is served from the Infinity Cache."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

SHAPES = [(15552, 1152, 1152, "res"), (15552, 1152, 4352, "res"), (15552, 3456, 1152, "bias"), (1376, 6144, 4096, ""), (1376, 4096, 4096, ""),
          (1376, 4096, 14336, ""), (1376, 28672, 4096, "swiglu"), (1024, 4096, 4096, ""), (2056, 1280, 1280, "res"), (2056, 3840, 1280, "bias"),
          (2056, 5120, 1280, "gelu"), (2056, 1280, 5120, "res"), (4112, 1280, 1280, "res"), (4112, 1280, 5120, "res"), (8224, 1280, 1280, "res"),
          (8224, 1280, 5120, "res"), (11000, 1280, 1280, "res"), (11000, 1280, 5120, "res")]
if len(sys.argv) > 1 and sys.argv[1] == "sweep":         # 256-tile counts around one round of the 256 CUs
    SHAPES = [(m, 4096, k, "") for k in (4096, 1280) for m in (1792, 2048, 2304, 3072, 4096, 4352, 4608, 5120, 6144)]
SELS = (0, 60, 70)
if len(sys.argv) > 1 and sys.argv[1] == "duo":           # the 128 x 256 "duo" experiment kernel (select 50) on the under-filled shapes
    SELS = (0, 50)
    SHAPES = [(1376, 4096, 4096, ""), (1376, 4096, 14336, ""), (1376, 6144, 4096, ""), (1024, 4096, 4096, ""), (2056, 1280, 5120, "res"), (2056, 1280, 1280, "res"), (4112, 1280, 5120, "res"), (4352, 4096, 4096, ""), (256, 12288, 4096, ""), (256, 22016, 4096, "")]
lib = _lib.lib()
_lib.lab()          # experiment kernels (csrc/lab/) register themselves with licv_gemm_select
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K, epi) in SHAPES:
    nbuf = max(2, -(-300 * 2 ** 20 // ((N * K + M * K + M * N) * 2)))
    As = [torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16) for _ in range(nbuf)]
    Ws = [(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16) for _ in range(nbuf)]
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    n_out = N // 2 if epi == "swiglu" else N
    res = torch.randn(M, n_out, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty(M, n_out, device="cuda", dtype=torch.bfloat16)
    kw = dict(res=dict(bias=bias, residual=res), bias=dict(bias=bias), gelu=dict(bias=bias, act="gelu"), swiglu=dict(swiglu=True)).get(epi, {})
    t, ref = {}, None
    for sel in SELS:
        lib.licv_gemm_select(sel)
        run = lambda i: ops.linear(As[i], Ws[i], out=out, **kw)
        for i in range(nbuf): run(i)
        if ref is None: ref = out.clone()
        same = torch.equal(ref, out)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nbuf): run(i)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / nbuf * 1e3)
        t[sel] = (best, same)
    lib.licv_gemm_select(0)
    print(f"{M:6d} {N:6d} {K:6d} {epi:7s} " + "  ".join(f"{s}: {t[s][0]:7.1f} us ({2.0 * M * N * K / t[s][0] / 1e6:5.0f} TF){'' if t[s][1] else ' DIFF'}" for s in t), flush=True)
    del As, Ws
