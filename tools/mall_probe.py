#!/usr/bin/env python3
"""Does a weight matrix that was just read (and so sits in the 256 MB Infinity Cache) stream faster into the M = 24 decode projection
than a cold one?  Per shape: the projection cold (distinct matrices cycled, > 600 MB), right after a full read of the same matrix by
another kernel (a torch reduction: 'prefetched'), after a read of its first 32 MiB only, and replayed on the same matrix (warm)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]
import torch
from licv import ops

g = torch.Generator(device="cuda").manual_seed(1)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3

for (M, N, K) in [(24, 12288, 4096), (24, 4096, 4096), (24, 22016, 4096), (24, 4096, 11008)]:
    nbuf = max(3, -(-700 * 2 ** 20 // (N * K * 2)))
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    for w in ws: ops.linear(a, w)
    torch.cuda.synchronize()
    res = {}
    for mode in ("cold", "prefetched", "prefetched 32 MiB", "warm"):
        best = 1e9
        for rep in range(3):
            ts = []
            for w in ws:
                if mode == "prefetched":
                    w.view(torch.int32).sum()
                elif mode == "prefetched 32 MiB":
                    w.view(torch.int32).view(-1)[: 8 * 2 ** 20].sum()
                elif mode == "warm":
                    ops.linear(a, w)
                ts.append(timed(lambda: ops.linear(a, w)))
            best = min(best, sum(ts) / len(ts))
        res[mode] = best
    print(f"{M:3d} {N:6d} {K:6d}  W {N * K * 2 / 2 ** 20:6.1f} MiB  " + "  ".join(f"{m}: {t:6.1f} us" for m, t in res.items()), flush=True)
    del ws
