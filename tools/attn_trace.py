#!/usr/bin/env python3
"""Phase timeline of attn_resident_k on the ViT shape (needs the library built with -DLICV_ATTN_TRACE): s_memtime stamps of the 8 waves
of workgroup 0 over their first items, printed as deltas in shader cycles."""
import ctypes
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L = _lib.lib()
fn = ctypes.CDLL(_lib.LIB_PATH if hasattr(_lib, "LIB_PATH") else str(ROOT / "licv-vqa_amd/licv/liblicv_hip.so")).licv_attn_debug_timestamps
fn.argtypes = [ctypes.c_void_p]
L.licv_attn_select(mode)
g = torch.Generator(device="cuda").manual_seed(0)
B, T, nh, hd = 264, 257, 16, 80
E = nh * hd
qkv = torch.randn(B * T, 3 * E, device="cuda", generator=g).to(torch.bfloat16)
run = lambda: ops.attention(qkv, qkv.view(-1)[E:], qkv.view(-1)[2 * E:], B, T, T, nh, nh, hd, T * 3 * E, 3 * E, T * 3 * E, 3 * E, hd ** -0.5, 0)
for _ in range(3):
    run()
buf = torch.zeros(8 * 128, dtype=torch.int64, device="cuda")
fn(buf.data_ptr())
run()
torch.cuda.synchronize()
fn(None)
t = buf.cpu().view(8, 128)
t0 = int(t[:, 0].min())
for w in range(8):
    ev = [int(x) - t0 for x in t[w] if int(x) != 0]
    print(f"wave {w}: first stamp +{ev[0]}; deltas:", " ".join(str(b - a) for a, b in zip(ev[:-1], ev[1:]))[:900])
