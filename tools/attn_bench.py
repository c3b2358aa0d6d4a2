#!/usr/bin/env python3
"""Attention kernel timing on the shapes of the headline step: ViT (264 images x 16 heads, 257 tokens, head dim 80, no mask),
language self-attention (8 x 32 heads, 800 tokens, head dim 128, causal + key padding), gated cross-attention (image mask)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import ops


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
B, T, nh, hd = 264, 257, 16, 80
E = nh * hd
qkv = torch.randn(B * T, 3 * E, device="cuda", generator=g).to(torch.bfloat16)
t = timed(lambda: ops.attention(qkv, qkv.view(-1)[E:], qkv.view(-1)[2 * E:], B, T, T, nh, nh, hd, T * 3 * E, 3 * E, T * 3 * E, 3 * E, hd ** -0.5, 0))
print(f"ViT        B {B} T {T} heads {nh} hd {hd}: {t:7.1f} us  {4.0 * B * nh * T * T * hd / t / 1e6:6.1f} TFLOP/s  "
      f"(HBM floor {(4 * B * T * E * 2) / 4.5e12 * 1e6:.0f} us at 4.5 TB/s)")
B, S, nh, hd = 8, 800, 32, 128
H = nh * hd
qkv = torch.randn(B * S, 3 * H, device="cuda", generator=g).to(torch.bfloat16)
valid = torch.ones(B, S, dtype=torch.int32, device="cuda")
valid[:, 760:] = 0
t = timed(lambda: ops.attention(qkv, qkv.view(-1)[H:], qkv.view(-1)[2 * H:], B, S, S, nh, nh, hd, S * 3 * H, 3 * H, S * 3 * H, 3 * H, hd ** -0.5, 1, key_valid=valid))
print(f"LM causal  B {B} S {S} heads {nh} hd {hd}: {t:7.1f} us  {2.0 * B * nh * S * S * hd / t / 1e6:6.1f} TFLOP/s (causal half)")

# SigLIP tower of the Idefics2 32-shot step: 132 images (one batch slice) x 16 heads, 980 / 14 = 70 x 70 = 4900 patch tokens would be the
# full-resolution case; the bench's images are 378 x 504: 27 x 36 = 972 tokens, head dim 72 (tiled kernel: the keys do not fit LDS)
B, T, nh, hd = 132, 972, 16, 72
E = nh * hd
qkv = torch.randn(B * T, 3 * E, device="cuda", generator=g).to(torch.bfloat16)
t = timed(lambda: ops.attention(qkv, qkv.view(-1)[E:], qkv.view(-1)[2 * E:], B, T, T, nh, nh, hd, T * 3 * E, 3 * E, T * 3 * E, 3 * E, hd ** -0.5, 0), n=5)
print(f"SigLIP     B {B} T {T} heads {nh} hd {hd}: {t:7.1f} us  {4.0 * B * nh * T * T * hd / t / 1e6:6.1f} TFLOP/s")
B, S, nh, nkv, hd = 4, 2900, 32, 8, 128
qd, kd = nh * hd, nkv * hd
ld = qd + 2 * kd
qkv = torch.randn(B * S, ld, device="cuda", generator=g).to(torch.bfloat16)
valid = torch.ones(B, S, dtype=torch.int32, device="cuda")
t = timed(lambda: ops.attention(qkv, qkv.view(-1)[qd:], qkv.view(-1)[qd + kd:], B, S, S, nh, nkv, hd, S * ld, ld, S * ld, ld, hd ** -0.5, 1, key_valid=valid), n=5)
print(f"Mistral    B {B} S {S} heads {nh}/{nkv} hd {hd}: {t:7.1f} us  {2.0 * B * nh * S * S * hd / t / 1e6:6.1f} TFLOP/s (causal half)")
