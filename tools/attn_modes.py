#!/usr/bin/env python3
"""Compare the schedules of the resident-K/V attention kernel (licv_attn_select modes) element by element."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

hd, S, Sq, B, nh = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (80, 257, 257, 3, 4)))
H = nh * hd
g = torch.Generator().manual_seed(52)
q = torch.randn(B, Sq, H, generator=g).to(torch.bfloat16).cuda()
kv = torch.randn(B, S, 2 * H, generator=g).to(torch.bfloat16).cuda()
outs = {}
for mode in (0, 4, 2, 1, 0):
    _lib.lib().licv_attn_select(mode)
    o = ops.attention(q, kv, kv.view(-1)[H:], B, Sq, S, nh, nh, hd, Sq * H, H, S * 2 * H, 2 * H, hd ** -0.5, 0).clone()
    if mode in outs:
        print("mode", mode, "repeat identical:", torch.equal(o, outs[mode]))
    outs[mode] = o
_lib.lib().licv_attn_select(0)
for m in (4, 2, 1):
    d = (outs[0].float() - outs[m].float()).abs().view(B, Sq, nh, hd)
    nz = (d > 0)
    print(f"mode 0 vs {m}: differing {int(nz.sum())} of {d.numel()}, max {float(d.max()):.3e}; by query-row mod 16: {nz.sum(dim=(0, 2, 3)).view(-1)[:Sq // 16 * 16].view(-1, 16).sum(0).tolist()}")
    print("   rows with differences:", nz.any(dim=3).any(dim=2).any(dim=0).nonzero().flatten().tolist()[:40])
    print("   dims with differences:", nz.any(dim=0).any(dim=0).any(dim=0).nonzero().flatten().tolist())
