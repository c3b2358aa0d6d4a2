#!/usr/bin/env python3
"""Same-process A/B of GEMM kernels on the headline shapes with the PLAIN epilogue (random bf16 operands, interleaved rounds:
cdna_hip_programming.md 5.4 rules 24/25), the platform library (F.linear -> hipBLASLt) timed beside them as select -1.
Usage: python tools/gemm_ab.py [--rounds 3] [--iters 8] [--shapes "M,N,K;..."] select [select ...]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from licv import _lib, ops  # noqa: E402

SHAPES = [(67848, 3840, 1280), (67848, 1280, 1280), (67848, 5120, 1280), (67848, 1280, 5120), (6400, 12288, 4096), (6400, 4096, 4096),
          (6400, 22016, 4096), (6400, 4096, 11008), (16896, 8192, 1280), (6400, 32002, 4096), (8192, 8192, 8192)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("select", nargs="*", type=int, default=[-1, 0, 22, 40])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--shapes", default="")
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.lab()          # experiment kernels (csrc/lab/) register themselves with licv_gemm_select
    shapes = SHAPES
    if args.shapes:
        shapes = [tuple(int(x) for x in s.split(",")) for s in args.shapes.split(";")]
    g = torch.Generator(device="cuda").manual_seed(1)
    geo = {s: 1.0 for s in args.select}
    for (M, N, K) in shapes:
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
        best, outs = {}, {}
        for _ in range(args.rounds):
            for sel in args.select:
                if sel >= 0:
                    lib.licv_gemm_select(sel)
                    fn = lambda: ops.linear(a, w)
                else:
                    fn = lambda: F.linear(a, w)
                for _ in range(2):
                    o = fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    o = fn()
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / args.iters * 1e-3
                best[sel] = max(best.get(sel, 0.0), 2.0 * M * N * K / t / 1e12)
                outs[sel] = o
        lib.licv_gemm_select(0)
        ours = [s for s in args.select if s >= 0]
        same = {s: bool(torch.equal(outs[s], outs[ours[0]])) for s in ours}
        for s in args.select:
            geo[s] *= best[s]
        print(f"{M:6d} {N:6d} {K:6d} " + "  ".join(f"{s}: {best[s]:7.1f}" for s in args.select)
              + "  equal=" + ",".join(str(int(same[s])) for s in ours), flush=True)
        del a, w, outs
    n = len(shapes)
    print("geomean       " + "  ".join(f"{s}: {geo[s] ** (1.0 / n):7.1f}" for s in args.select), flush=True)


if __name__ == "__main__":
    main()
