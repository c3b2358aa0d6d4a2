#!/usr/bin/env python3
"""The two small backward kernels of the student pass at its real shapes, old form against new (licv_backward_option): attention backward
(B = 8, S = 32, 32 heads x 128; operands in global memory | staged in LDS) and RMSNorm backward (256 rows x 4096, fp32 stream;
one wave per row | four)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)

def timed(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

B, S, nh, hd = 8, 32, 32, 128
for nkv in (32, 8):
    qd, kd = nh * hd, nkv * hd
    ldq = qd + 2 * kd
    qkv = torch.randn(B * S, ldq, device="cuda", generator=g).to(torch.bfloat16)
    dout = (torch.randn(B * S, qd, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    kval = torch.ones(B, S, dtype=torch.int32, device="cuda")
    dq = torch.zeros(B * S, qd, dtype=torch.bfloat16, device="cuda")
    dkv = torch.zeros(B * S, 2 * qd, dtype=torch.bfloat16, device="cuda")
    def run():
        ops.attention_bwd_small(qkv, qkv.view(-1)[qd:], qkv.view(-1)[qd + kd:], dout, B, S, S, nh, nkv, hd, S * ldq, ldq, S * ldq, ldq, hd ** -0.5, 1,
                                dq, S * qd, qd, dk=dkv, dv=dkv.view(-1)[qd:], dkv_bs=S * 2 * qd, dkv_rs=2 * qd, key_valid=kval)
    t = {}
    for staged in (0, 2, 1):
        lib.licv_backward_option(0, staged)
        t[staged] = timed(run)
    print(f"attn_bwd_small B={B} S={S} {nh}q/{nkv}kv x {hd}: global operands {t[0]:6.1f} us   staged in LDS, 256 lanes {t[2]:6.1f} us   staged, 1024 lanes {t[1]:6.1f} us", flush=True)
for rows, dim, dt in ((256, 4096, torch.float32), (256, 4096, torch.bfloat16), (768, 4096, torch.float32), (256 * 32, 128, torch.bfloat16)):
    x = torch.randn(rows, dim, device="cuda", generator=g).to(dt)
    w = torch.ones(dim, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(rows, dim, device="cuda", generator=g).to(dt)
    dx = torch.zeros(rows, dim, device="cuda", dtype=dt)
    t = {}
    for wide in (0, 1):
        lib.licv_backward_option(1, wide)
        t[wide] = timed(lambda: ops.rmsnorm_bwd(x, w, dy, dx, 1e-6, accumulate=False, flavour=1))
    print(f"rmsnorm_bwd {rows} x {dim} {str(dt)[6:]}: one wave per row {t[0]:6.1f} us   four waves per row {t[1]:6.1f} us", flush=True)
lib.licv_backward_option(0, 1); lib.licv_backward_option(1, 1)
