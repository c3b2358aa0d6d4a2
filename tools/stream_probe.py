#!/usr/bin/env python3
"""HBM rate of the weight-stream access shapes (licv_probe_weight_stream), cold: 7 x 96 MiB matrices cycled."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib
lib = _lib.lab()          # roofline probes live in liblicv_hip_lab.so
N, K = 12288, 4096
ws = [torch.randn(N, K, device="cuda").to(torch.bfloat16) for _ in range(7)]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for splits in (1, 2, 4, 8):
    for shape in (0, 1, 2, 3):
        for depth in (4, 8, 16):
            def run(w): assert lib.licv_probe_weight_stream(w.data_ptr(), K, N, K, splits, shape, depth, sink.data_ptr(), st) == 0
            for w in ws: run(w)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for w in ws: run(w)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / len(ws) * 1e3)
            print(f"splits {splits} ({N // 64 * splits:5d} workgroups) shape {shape} depth {depth:2d}: {best:6.1f} us  {N * K * 2 / best / 1e6:5.2f} TB/s", flush=True)

# the same matrix as LDS-DMA pieces: pieces outstanding per wave
for splits in (1, 2, 4):
    for depth in (4, 8, 16, 32, 48):
        def run(w): assert lib.licv_probe_lds_dma_stream(w.data_ptr(), K, N, K, splits, depth, st) == 0
        for w in ws: run(w)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for w in ws: run(w)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / len(ws) * 1e3)
        print(f"LDS-DMA splits {splits} ({N // 64 * splits:5d} workgroups) depth {depth:2d}: {best:6.1f} us  {N * K * 2 / best / 1e6:5.2f} TB/s", flush=True)
