#!/usr/bin/env python3
"""Where one mid-loop K stage of the ping-pong GEMM spends its cycles, per wave (diagnostic build, licv_gemm_select(13)):
s_memtime stamps around fragment reads / DMA issue / counted vmcnt / lgkmcnt / barrier / MFMA issue / barrier."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

lib = _lib.lib()

_lib.lab()          # experiment kernels (csrc/lab/) register themselves with licv_gemm_select
SEL = int(sys.argv[1]) if len(sys.argv) > 1 else 13        # 13: ping-pong kernel, 25: lean kernel (stamped builds)
names = ["ds_read issue", "DMA issue", "vmcnt wait", "lgkmcnt wait", "barrier 1", "MFMA issue", "barrier 2"]
for (M, N, K) in [(6400, 12288, 4096), (67848, 3840, 1280), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    nt = ((M + 255) // 256) * ((N + 255) // 256)
    ts = torch.zeros(nt * 64, dtype=torch.int64, device="cuda")
    lib.licv_gemm_select(SEL)
    for _ in range(3):
        ops.linear(a, w)
    assert lib.licv_gemm_debug_timestamps(ts.data_ptr()) == 0
    ops.linear(a, w)
    torch.cuda.synchronize()
    assert lib.licv_gemm_debug_timestamps(None) == 0
    lib.licv_gemm_select(0)
    t = ts.view(nt, 8, 8).cpu().double()
    seg = t[:, :, 1:] - t[:, :, :-1]                       # (tile, wave, 7)
    ok = (t[:, :, 0] > 0).all(dim=1)
    seg = seg[ok]
    print(f"{M} x {N} x {K}: {int(ok.sum())} tiles; cycles per segment (median over tiles), leading waves 0-3 | trailing waves 4-7")
    for i, nme in enumerate(names):
        lead, trail = seg[:, :4, i].median(), seg[:, 4:, i].median()
        print(f"   {nme:14s} {float(lead):8.0f} | {float(trail):8.0f}")
    rel = (t - t[:, :1, :1])[ok]                            # every stamp relative to wave 0's load-phase start of the same tile
    print("   timeline (median over tiles, cycles after wave 0's stamp 0); columns = stamps 0..7")
    for w in range(8):
        print(f"     wave {w}: " + " ".join(f"{float(rel[:, w, i].median()):7.0f}" for i in range(8)))
    ld = (t[:, :, 4] - t[:, :, 0])[ok]
    print(f"   load phase (stamps 0->4) per tile: median wave {float(ld.median()):.0f}, slowest of the 4 leaders {float(ld[:, :4].max(dim=1).values.median()):.0f}, "
          f"slowest of the 4 trailers {float(ld[:, 4:].max(dim=1).values.median()):.0f}; p90 of a wave {float(ld.flatten().quantile(0.9)):.0f}")
    tot = (t[:, :, 7] - t[:, :, 0])[ok]
    for nme, x in (("stage total", tot.flatten()), ("vmcnt wait", seg[:, :, 2].flatten()), ("barrier 1", seg[:, :, 4].flatten()), ("barrier 2", seg[:, :, 6].flatten())):
        print(f"   {nme:12s} distribution over tiles x waves: mean {float(x.mean()):.0f}  p50 {float(x.median()):.0f}  p90 {float(x.quantile(0.9)):.0f}  "
              f"p99 {float(x.quantile(0.99)):.0f}  max {float(x.max()):.0f}")
    print(f"   stage total    {float(tot[:, :4].median()):8.0f} | {float(tot[:, 4:].median()):8.0f}   (MFMA-bound would be 2 x 512 = 1024 per SIMD)")
