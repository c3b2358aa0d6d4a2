#!/usr/bin/env python3
"""Effective shader clock a GEMM kernel holds (MI355X_MICROARCH.md 'DVFS give-back': GRBM_GUI_ACTIVE / 8 / kernel wall time).

  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 tools/clock_probe.py run
  python3 tools/clock_probe.py parse DIR
(the same `run` under `--pmc FETCH_SIZE` or `--pmc WRITE_SIZE` gives the beyond-L2 traffic per launch of each kernel; `parse` prints it)
CLOCK_PROBE_SHAPE=M,N,K overrides the shape.

`run` launches, back to back on random operands, the vendor-library GEMM (torch F.linear) and this library's kernels (select
values below) on a shape long enough for the quotient to be meaningful (>= 1.5 ms per dispatch)."""
import csv, sys
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "licv-vqa_amd")]

import os
SHAPE = tuple(int(x) for x in os.environ.get("CLOCK_PROBE_SHAPE", "16384,8192,8192").split(","))
SELECTS = [int(x) for x in os.environ.get("CLOCK_PROBE_SELECTS", "6,22,20").split(",")]


def run():
    import torch
    from licv import _lib, ops
    lib = _lib.lib()
    _lib.lab()          # experiment kernels (csrc/lab/) register themselves with licv_gemm_select
    M, N, K = SHAPE
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    for rounds in range(3):
        for _ in range(12):
            torch.nn.functional.linear(a, w)
        for sel in SELECTS:
            lib.licv_gemm_select(sel)
            for _ in range(12):
                ops.linear(a, w)
        lib.licv_gemm_select(0)
    torch.cuda.synchronize()


def parse(d):
    dur, name = {}, {}
    for f in Path(d).rglob("*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            name[r["Dispatch_Id"]] = r["Kernel_Name"]
    acc = defaultdict(list)
    traffic = defaultdict(list)
    for f in Path(d).rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):          # KiB; FETCH_SIZE x2 on gfx950 (guide, HBM section)
                traffic[(r["Kernel_Name"][:90], r["Counter_Name"])].append(float(r["Counter_Value"]) * 1024.0 * (2.0 if r["Counter_Name"] == "FETCH_SIZE" else 1.0))
                continue
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            t = dur.get(r["Dispatch_Id"])
            if t is None and "Start_Timestamp" in r:
                t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            if not t or t < 3e-4:
                continue
            acc[r["Kernel_Name"][:90]].append((float(r["Counter_Value"]) / 8.0 / t, t))
    M, N, K = SHAPE
    for k, v in acc.items():
        v = v[len(v) // 3:]                                  # drop the first third (clock still settling)
        clk = sorted(x[0] for x in v)[len(v) // 2]
        t = sorted(x[1] for x in v)[len(v) // 2]
        tf = 2.0 * M * N * K / t / 1e12
        print(f"{k[:70]:70s} n={len(v):3d}  clock {clk / 1e9:5.2f} GHz  {t * 1e3:6.3f} ms  {tf:7.1f} TF  "
              f"= {tf / (2500.0 * clk / 2.4e9):.3f} of the MFMA peak AT THAT CLOCK")


    for (k, c), v in sorted(traffic.items()):
        if max(v) > 1e6:
            print(f"{k[:70]:70s} {c}: {sorted(v)[len(v) // 2] / 1e9:.3f} GB per launch (median of {len(v)}); operands A+W {2.0 * (M + N) * K / 1e9:.3f} GB, C {2.0 * M * N / 1e9:.3f} GB")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else parse(sys.argv[2])
