#!/usr/bin/env python3
"""One-line digest of bench.py output files: median ms per step, questions/s, the GEMM roofline numbers.
usage: bench_line.py <bench.json.log> [...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(f"{f}: {d['ms_per_step_median']:.1f} ms median, {d['value']:.2f} {d['unit']}; GEMM {r.get('achieved') or 0:.0f} TF/s "
          f"(frac {r.get('frac') or 0:.3f}), avg launch {r.get('avg_launch_us') or 0:.0f} us"
          + (f" (start-to-end {r['avg_launch_us_start_to_end']:.0f} us, {r.get('batch_streams')} slices)" if r.get("avg_launch_us_start_to_end") else "")
          + f", GEMM share {r.get('gemm_share_of_step') or 0:.3f}")
