#!/usr/bin/env python3
"""A/B microbenchmark of the GEMM kernels on the real shapes of the headline forward (random operands, interleaved rounds in one
process: cdna_hip_programming.md §5.4 rules 24/25).  Usage:  python tools/gemm_bench.py [--rounds 3] [--iters 10] [select ...]
Each `select` is a licv_gemm_select() value (0 = default dispatch, 6 = ping-pong, 20 = flow kernel where eligible, 8 = persistent).
Prints TFLOP/s per (shape, epilogue, kernel) and whether every kernel's output equals the first one's bit for bit."""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch  # noqa: E402

from licv import _lib, ops  # noqa: E402

# (M, N, K, epilogue) of the Idefics-9B 32-shot bs8 forward; N of swiglu shapes is the packed 2I
SHAPES = [
    (67848, 3840, 1280, "bias"), (67848, 1280, 1280, "bias+res"), (67848, 5120, 1280, "bias+gelu"), (67848, 1280, 5120, "bias+res"),
    (6400, 12288, 4096, "plain"), (6400, 4096, 4096, "res32"), (6400, 22016, 4096, "swiglu"), (6400, 4096, 11008, "res32"),
    (16896, 8192, 1280, "plain"), (6400, 32002, 4096, "plain"),
]


def run(M, N, K, epi, a, w, bias, res16, res32):
    kw = {}
    if "bias" in epi:
        kw["bias"] = bias
    if "gelu" in epi:
        kw["act"] = "gelu"
    if epi == "swiglu":
        kw["swiglu"] = True
    if "res32" in epi:
        return ops.linear(a, w, residual=res32, out=res32.clone(), **kw)
    if "res" in epi:
        return ops.linear(a, w, residual=res16, out=res16.clone(), **kw)
    return ops.linear(a, w, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("select", nargs="*", type=int, default=[6, 20])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--shapes", default="")
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.lab()          # experiment kernels (csrc/lab/) register themselves with licv_gemm_select
    shapes = SHAPES
    if args.shapes:
        shapes = [tuple(int(x) if x.isdigit() else x for x in s.split(",")) for s in args.shapes.split(";")]
    g = torch.Generator(device="cuda").manual_seed(1)
    for (M, N, K, epi) in shapes:
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
        n_out = N // 2 if epi == "swiglu" else N
        bias = (torch.randn(N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
        res16 = torch.randn(M, n_out, device="cuda", generator=g).to(torch.bfloat16) if "res" in epi and "32" not in epi else None
        res32 = torch.randn(M, n_out, device="cuda", generator=g) if "res32" in epi else None
        best, outs = {}, {}
        variants = args.select
        for _ in range(args.rounds):
            for sel in variants:
                lib.licv_gemm_select(sel)
                for _ in range(2):
                    o = run(M, N, K, epi, a, w, bias, res16, res32)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if "res" in epi:                       # the clone is outside what is measured: time the kernel alone through ops' profiler
                    prof = []
                    ops.set_profiler(prof)
                    for _ in range(args.iters):
                        o = run(M, N, K, epi, a, w, bias, res16, res32)
                    torch.cuda.synchronize()
                    ops.set_profiler(None)
                    t = sum(r[1].elapsed_time(r[2]) for r in prof) / len(prof) * 1e-3
                else:
                    e0.record()
                    for _ in range(args.iters):
                        o = run(M, N, K, epi, a, w, bias, res16, res32)
                    e1.record()
                    torch.cuda.synchronize()
                    t = e0.elapsed_time(e1) / args.iters * 1e-3
                best[sel] = max(best.get(sel, 0.0), 2.0 * M * N * K / t / 1e12)
                outs[sel] = o
        lib.licv_gemm_select(0)
        first = variants[0]
        same = {s: bool(torch.equal(outs[s], outs[first])) for s in variants}
        print(f"{M:6d} {N:6d} {K:6d} {epi:10s} " + "  ".join(f"{s}: {best[s]:7.1f} TF" for s in variants)
              + "  equal=" + ",".join(str(int(same[s])) for s in variants), flush=True)
        del a, w, outs


if __name__ == "__main__":
    main()
