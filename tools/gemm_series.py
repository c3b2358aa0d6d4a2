#!/usr/bin/env python3
"""Time series of the lean GEMM kernel's main loop (diagnostic build, licv_gemm_select(27)): shader cycles per 32-deep K stage
as a function of the position in the tile's K sweep, averaged over tiles and waves.
usage: gemm_series.py [M,N,K ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

lib = _lib.lib()
shapes = [(8192, 8192, 8192), (6400, 12288, 4096), (67848, 3840, 1280)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    nt = ((M + 255) // 256) * ((N + 255) // 256)
    ns = K // 32
    stride = (ns + 63) // 64
    ts = torch.zeros(nt * 8 * 64, dtype=torch.int64, device="cuda")
    lib.licv_gemm_select(27)
    for _ in range(20):
        ops.linear(a, w)
    assert lib.licv_gemm_debug_timestamps(ts.data_ptr()) == 0
    ops.linear(a, w)
    torch.cuda.synchronize()
    assert lib.licv_gemm_debug_timestamps(None) == 0
    lib.licv_gemm_select(0)
    t = ts.view(nt, 8, 64).cpu()
    npts = min((ns + stride - 1) // stride + 1, 64)             # stamps at stages 0, stride, ..., and one after the last stage
    t = t[:, :, :npts]
    d = ((t[:, :, 1:] - t[:, :, :-1]) & 0xFFFFFFFF).double()   # 32-bit counter differences
    span = [stride] * (npts - 2) + [max(ns - stride * (npts - 2), 1)]
    per = d / torch.tensor(span, dtype=torch.float64)
    mean = per.mean(dim=(0, 1))
    print(f"{M} x {N} x {K}: {nt} tiles, {ns} stages, one stamp per {stride} stage(s); cycles per stage along the K sweep (mean over tiles and waves)")
    print("   " + " ".join(f"{float(x):5.0f}" for x in mean))
    print(f"   whole loop: mean {float(d.sum(dim=2).mean()) / ns:.0f} cycles per stage; middle half {float(per[:, :, npts // 4: 3 * npts // 4].mean()):.0f}; "
          f"p10 / p50 / p90 of a stamp interval {float(per.flatten().quantile(0.1)):.0f} / {float(per.flatten().median()):.0f} / {float(per.flatten().quantile(0.9)):.0f}")
