#!/usr/bin/env python3
"""Per-CU operand bandwidth from an L2 / Infinity-Cache resident buffer, by path (licv_probe_l2_ingest, lab library): VGPR loads,
LDS-DMA, and both at once.  One workgroup per CU (256) and two per CU (512), buffers of 1 / 2 / 8 MiB (L2 resident per XCD: 4 MiB)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib
lab = _lib.lab()
st = torch.cuda.current_stream().cuda_stream
for mib in (1, 2, 8):
    buf = torch.randint(0, 255, (mib << 20,), dtype=torch.uint8, device="cuda")
    for blocks in (256, 512):
        for mode, name in ((0, "VGPR loads"), (1, "LDS-DMA"), (2, "both (2 + 2 waves)")):
            reps = max(1, 64 // mib)
            run = lambda: lab.licv_probe_l2_ingest(buf.data_ptr(), buf.numel(), reps, mode, blocks, None, st)
            assert run() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 5 * 1e-3
            per_wg = buf.numel() * reps * (2 if mode == 2 else 1)
            print(f"{mib} MiB buffer, {blocks} workgroups, {name:20s}: {t * 1e6:8.1f} us  {per_wg / t / 1e9:7.1f} GB/s per workgroup  "
                  f"{per_wg * blocks / t / 1e12:6.2f} TB/s chip", flush=True)
