#!/usr/bin/env python3
"""The ViT attention launch of the headline step alone (264 images x 16 heads, 257 tokens, head dim 80), a few times: the program
rocprofv3 --pmc runs for the wave-state / LDS / L2 counters of attn_resident_k.  argv[1] = licv_attn_select mode (default 0)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
_lib.lib().licv_attn_select(mode)
g = torch.Generator(device="cuda").manual_seed(0)
B, T, nh, hd = 264, 257, 16, 80
E = nh * hd
qkv = torch.randn(B * T, 3 * E, device="cuda", generator=g).to(torch.bfloat16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(n):
    e0.record()
    o = ops.attention(qkv, qkv.view(-1)[E:], qkv.view(-1)[2 * E:], B, T, T, nh, nh, hd, T * 3 * E, 3 * E, T * 3 * E, 3 * E, hd ** -0.5, 0)
    e1.record()
    torch.cuda.synchronize()
    print(f"mode {mode} launch {it}: {e0.elapsed_time(e1) * 1e3:.1f} us")
