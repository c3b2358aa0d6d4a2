#!/usr/bin/env python3
"""The 256 x 128 tall kernel (licv_gemm_select 71; 129-256 rows) against the 128-tile mid kernel on the M = 256 weight-streaming
shapes of the student pass and of generate's prefill, COLD (distinct matrices cycled, as tools/split_sweep.py), under the plan's own
split-K choice (sp0) and forced counts (knob 5): the data the plan's tall constants are fitted to.  Also checks that at every forced
count the two producers give the same bits."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import _lib, ops

SHAPES = [(256, 12288, 4096, ""), (256, 4096, 4096, ""), (256, 22016, 4096, "swiglu"), (256, 4096, 11008, ""), (256, 32002, 4096, ""),
          (256, 8192, 4096, ""), (256, 2048, 4096, ""), (256, 1024, 4096, ""), (200, 4096, 4096, ""), (256, 4096, 1280, ""), (256, 28672, 4096, "swiglu"),
          (256, 4096, 14336, ""), (256, 6144, 4096, "")]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    SHAPES = SHAPES[:5]
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K, epi) in SHAPES:
    nbuf = max(2, -(-640 * 2 ** 20 // (N * K * 2)))
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16) for _ in range(nbuf)]
    kw = dict(swiglu=True) if epi == "swiglu" else {}
    lines = {}
    outs = {}
    for tall in (0, 1):
        lib.licv_gemm_select(71 if tall else 70)             # forced: the tall kernel wherever it can run / the mid kernel
        line = f"{M:4d} {N:6d} {K:6d} {epi:6s} {'tall' if tall else 'mid '}"
        for sp in (0, 1, 2, 3, 4, 6, 8, 12, 16):
            if sp > 1 and K // 64 // sp < 2:
                continue
            lib.licv_gemm_experiment(5, sp)
            try:
                for w in ws: ops.linear(a, w, **kw)
            except Exception as e:
                line += f"  sp{sp}: err"; continue
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for w in ws: ops.linear(a, w, **kw)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / nbuf * 1e3)
            line += f"  sp{sp}: {best:5.1f}"
            if sp:
                outs[(tall, sp)] = ops.linear(a, ws[0], **kw).clone()
        lib.licv_gemm_experiment(5, 0)
        print(line, flush=True)
    lib.licv_gemm_select(0)
    bad = [sp for (t, sp) in outs if t == 0 and (1, sp) in outs and not torch.equal(outs[(0, sp)], outs[(1, sp)])]
    print(f"     bits: {'identical at every forced count' if not bad else 'DIFFER at sp ' + str(bad)}", flush=True)
    del ws
